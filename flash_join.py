"""Drop-in for the reference's `flash_join` extension module (PYBIND11_MODULE, hash_join.cpp:598).

    import flash_join
    flash_join.initialize()
    count, core_seconds = flash_join.hash_join_count_radix(build_keys, build_values, probe_keys)

All names, keyword arguments and return shapes follow the reference; execution is on MI355X.
"""
from flash_hash_join_amd.api import *  # noqa: F401,F403
from flash_hash_join_amd.api import last_timings, join_device  # noqa: F401

__doc__ = "A high-performance hash join library with adaptive and explicit strategies."  # hash_join.cpp:600
