"""unimatch_amd: the UniMatch (GMFlow / GMStereo / GMDepth) global-matching hot path on AMD MI355X (gfx950).

``UniMatch`` is a drop-in for ``unimatch.unimatch.UniMatch`` of autonomousvision/unimatch; its attention,
matching, propagation and cost-volume layers run on hand-written HIP kernels behind the C ABI in
``include/unimatch_hip.h``.
"""
from .model import UniMatch  # noqa: F401
from .streams import ConcurrentUniMatch  # noqa: F401

__all__ = ['UniMatch', 'ConcurrentUniMatch']
__version__ = '0.1.0'
