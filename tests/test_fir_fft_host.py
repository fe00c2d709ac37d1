"""The arithmetic core of the HRTF stage by overlap-save FFT (iac_amd/csrc/render_fir_fft.hpp) is __host__ __device__:
tests/fft_host/fft_host_check.cpp runs it lane by lane on the CPU (64 "lanes", an array as the wave's LDS scratch) and
checks the staged 1024-point transform against a float64 DFT, inverse(forward(x)) = N x, and whole overlap-save hops
(pair packing, U / V accumulation with the host-built tables, mirror exchange, inverse) against a float64 direct
convolution for 16 / 9 / 1 / 12 channels.  What this pins without a GPU: the index algebra of the three register stages,
both LDS exchange layouts, the bin order of the tables, the mirror positions.  (The same header compiled for gfx950 is
checked on the GPU by tests/test_gpu_fir.py.)"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_fft_core_on_the_host(tmp_path):
    exe = os.path.join(str(tmp_path), "fft_host_check")
    src = os.path.join(ROOT, "tests", "fft_host", "fft_host_check.cpp")
    cc = CLANG if os.path.exists(CLANG) else "clang++"
    subprocess.check_call([cc, "-O2", "-std=c++17", "-o", exe, src, "-lm"])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    out = p.stdout
    assert "bins covered once 1024 / 1024" in out and out.strip().endswith("OK")
    errs = [float(m) for m in re.findall(r"hop M=\d+ taps=\d+: max \|err\| ([0-9.e+-]+)", out)]
    assert len(errs) == 4 and max(errs) < 2.0 ** -19      # a quarter of the stated float tolerance 2^-17
