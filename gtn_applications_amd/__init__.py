"""MI355X-native differentiable-WFST loss engine: drop-in criteria for the hot path of
facebookresearch/gtn_applications (criterions/{ctc,asg,stc,transducer}.py), backed by hand-written
gfx950 HIP kernels behind the C ABI of include/wfl.h (libwfl.so).  See DESIGN.md."""
from . import _native  # noqa: F401  (fails loudly if libwfl.so is missing)
from . import graph  # noqa: F401

__all__ = ["graph", "criterions", "engine"]
