"""gcm_filters_amd: the gcm-filters iterated-Laplacian diffusion filter on AMD MI355X (gfx950).

Drop-in for the hot path of ``gcm_filters`` (``Filter.apply`` / ``apply_to_vector`` and the per-grid
Laplacians); same names as ``gcm_filters/__init__.py``.  The numerics run in hand-written HIP kernels
behind the C ABI of ``include/gcmf.h`` -- there is no CPU fallback.
"""
__version__ = "0.1.0"

from .filter import Filter, FilterShape
from .kernels import GridType, required_grid_vars

__all__ = ["Filter", "FilterShape", "GridType", "required_grid_vars"]
