"""MI355X-native SEDT forward/backward path (HIP kernels behind the reference's module API)."""
from .lib import F32, BF16  # noqa: F401
