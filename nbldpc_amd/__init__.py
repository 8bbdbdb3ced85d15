"""nbldpc_amd -- MI355X-native batched non-binary LDPC decode path (drop-in for YongonY/NBLDPC's
CNBLDPC::Decoding hot path).  The compute lives in csrc/libnbldpc_hip.so (hand-written HIP for gfx950,
C ABI in include/nbldpc.h); this package is the thin Python plumbing used by tests and bench.py."""
from . import datafiles  # noqa: F401
from .binding import (Code, Decoder, NblError, load_library, LIB_PATH, EXPORTS,  # noqa: F401
                      METHOD_BP, METHOD_EMS, METHOD_TEMS)
