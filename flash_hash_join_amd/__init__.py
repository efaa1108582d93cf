"""flash_hash_join_amd -- MI355X-native (gfx950) hash join behind the `flash_join` API.

`import flash_join` (repo-root shim) gives the reference's module surface; this package holds the
HIP kernels + C ABI (csrc/, include/flashjoin.h), the host-side mirror of the reference interface
(api.py), the synthetic generators (datagen.py) and the multi-GPU driver (distributed.py).
"""
from .api import *  # noqa: F401,F403
from .api import REFERENCE_EXPORTS, ALIASES, last_timings, join_device, context  # noqa: F401

__version__ = "0.2.0"
