"""trew_amd -- MI355X-native telomeric-repeat scanner (hot path of Chemical118/TREW).

The product is the HIP shared library (trew_amd/csrc -> trew_amd/lib/libtrew_hip.so,
C ABI in include/trew_hip.h) and the C++ `trew` host built on it.  This package
only binds the C ABI for tests and benchmarks.
"""
from . import capi  # noqa: F401
from .capi import (  # noqa: F401
    FLAG_NO_FILTER,
    FLAG_DEBUG_POISON_LDS,
    FLAG_NO_TIMING,
    FLAG_TRACK_PRESSURE,
    FLAG_COMPAT_G1,
    FLAG_DEBUG_NO_GROUP,
    FLAG_DEBUG_NO_JOINT,
    FLAG_DEBUG_WIDE_NO_WAIT,
    MODE_LONG,
    MODE_PAIR,
    MODE_SEGMENT,
    MODE_SHORT,
    TABLE_NAMES,
    TrewHip,
    TrewHipError,
    k_mer_check,
    pack_reads,
)
