import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

    class G:
        manifest = json.load(open(os.path.join(g, "manifest.json")))

        def __init__(self):
            self._c = {}

        def npz(self, name):
            if name not in self._c:
                self._c[name] = np.load(os.path.join(g, name + ".npz"))
            return self._c[name]

    return G()
