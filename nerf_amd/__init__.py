"""nerf_amd -- MI355X-native NeRF render-and-train hot path (libnerfhip.so) and its host-side mirror of
the reference's `nerfBlock` interface.  See DESIGN.md and include/nerfhip.h."""
from .config import NerfConfig  # noqa: F401
from .block import NerfBlock  # noqa: F401
from ._lib import NerfHipError  # noqa: F401
