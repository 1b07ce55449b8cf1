"""pearray_amd -- MI355X-native backend for PearRay's `direct` (spectral path tracing) hot path.

Only what the path needs lives here: `csrc/` (HIP kernels + C ABI `libprgpu.so`), `scene` (flat scene
assembly) and `backend` (ctypes host mirror).  Importing the package does not load the GPU library;
`backend.RenderContext` does, and raises if it has not been built.
"""
from . import _cabi  # noqa: F401
from ._cabi import (FILTER_BLOCK, FILTER_GAUSSIAN, FILTER_MITCHELL, FILTER_TRIANGLE, MAPPER_CIE, MAPPER_CIE_Y, MAPPER_RANDOM,  # noqa: F401
                    MAPPER_SPD_CMIS, MAPPER_SPD_HERO, MIS_BALANCE, MIS_POWER, SAMPLER_MJITT, SAMPLER_RANDOM,
                    SAMPLER_SOBOL)
