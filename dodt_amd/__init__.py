"""dodt_amd -- MI355X-native hot path of DODT/AVOD behind the avod.core names.

Host side is plain Python + numpy calling hand-written HIP kernels for gfx950
through the C-ABI in include/dodt_hip.h (ctypes).  No PyTorch, no CPU fallback:
every compute entry point raises if libdodt_hip.so is missing.
"""
