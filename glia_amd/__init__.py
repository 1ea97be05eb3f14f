"""glia_amd -- MI355X-native hierarchical-merge-tree (HMT) hot path behind GLIA's operator interface.

The compute lives in glia_amd/libglia_hmt.so (hand-written HIP for gfx950, C ABI in include/glia_hmt.h);
this package is the thin host-side mirror used by tests and bench.py.  There is no CPU fallback: importing
`glia_amd.hmt` fails loudly when the library is missing.
"""
from . import hmt  # noqa: F401
