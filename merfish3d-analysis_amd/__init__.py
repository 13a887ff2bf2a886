"""merfish3d-analysis_amd - MI355X-native per-tile preprocess -> decode hot path.

Drop-in for the hot path of QI2lab/merfish3d-analysis (RLGC deconvolution,
fiducial registration, PixelDecoder codebook matching): same operator / utility
names and signatures on top of hand-written HIP kernels for gfx950
(``csrc/libm3d_hip.so``, C ABI in ``include/m3d_hip.h``).

The directory name carries a hyphen, so import it either through the alias
module at the repository root (``import merfish3d_analysis_amd``) or with
``importlib.import_module("merfish3d-analysis_amd")``.
"""

__version__ = "0.1.0"

__all__ = ["__version__"]
