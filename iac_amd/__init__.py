"""iac_amd — MI355X-native IAMF post-decode renderer.

The product is the C-ABI shared library ``iac_amd/lib/libiamf_hip.so`` (HIP kernels for gfx950
plus a plain-C host side; headers in ``include/``).  This Python package is only a ctypes
binding used by the tests and bench.py: it holds no arithmetic and no CPU fallback.  Importing
it never touches ``oracle/``.
"""
from .hipabi import (  # noqa: F401
    FMT_F32,
    FMT_S16,
    FMT_S24,
    FMT_S32,
    KIND_DMX,
    KIND_FIR,
    KIND_H2M,
    KIND_M2M,
    PROJ_AUTO,
    PROJ_EXACT,
    PROJ_MFMA,
    SS,
    Batch,
    BatchConfig,
    DemixConfig,
    DemixFrame,
    DemixState,
    DmxFrame,
    DmxState,
    RenderArgs,
    Resampler,
    dmx_matrix,
    fir_matrix,
    IamfHipError,
    LpcmLayout,
    LpcmInput,
    Matrix,
    build,
    get_h2m_matrix,
    get_m2m_matrix,
    layout_channels,
    lib,
    lib_path,
    lpcm_unpack,
)
