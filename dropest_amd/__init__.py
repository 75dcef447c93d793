"""dropest_amd -- MI355X-native Estimation hot path of dropEst (HIP kernels behind a C-ABI).

The product path lives in csrc/ (HIP, gfx950) and is reached through include/dropest_amd.h; this Python
package is only the ctypes plumbing used by the tests and bench.py.  There is no CPU fallback.
"""
from .build import build  # noqa: F401
from .capi import (BARCODES_CONST, BARCODES_INDROP, ESCAPE, MERGE_NONE, MERGE_REAL_BARCODES, NO_GENE, Context,  # noqa: F401
                   DeviceArrays, DropestError, lib, pack_seq, unpack_code)
