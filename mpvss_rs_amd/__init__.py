"""mpvss_rs_amd -- MI355X-native batch group-exponentiation engine for the mpvss-rs PVSS hot path.

The product is the C-ABI shared library `libmpvss_hip.so` (sources in csrc/, interface in
include/mpvss_hip.h); this package only carries the ctypes binding used by the tests, bench.py
and __graft_entry__.py.
"""
from .capi import Engine, EngineError, load_library, LIB_PATH, EXPORTED_SYMBOLS  # noqa: F401
