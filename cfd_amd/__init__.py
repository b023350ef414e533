"""cfd_amd -- MI355X-native hot path of the CFD surrogate (annguyenphanhuu/CFD).

Host side is Python; every arithmetic op of the path runs in hand-written gfx950 HIP
kernels reached through the C-ABI library `cfd_amd/lib/libcfd_amd.so`
(declarations: include/cfd_amd.h).  There is no CPU or eager-PyTorch fallback: ops raise
if the library is missing.
"""
__version__ = "0.1.0"
