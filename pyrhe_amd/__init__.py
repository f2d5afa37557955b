"""pyrhe_amd -- MI355X-native streaming randomized-trace sketch engine behind PyRHE's model API.

Only the hot path lives here: the per-jackknife-block sketch of ``StreamingBase`` as hand-written
gfx950 HIP kernels behind a C ABI (``include/pyrhe_sketch.h``), the host-side mirror of the
reference's streaming driver / estimator, and the multi-GPU block sharding.
"""
from .build import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
__version__ = "0.1.0"
