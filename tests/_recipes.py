"""Input recipes of SURVEY.md Appendix D (known answers recorded from the reference's own headers)."""
import numpy as np


def _lcg(n, s0=12345, a=1664525, c=1013904223):
    """32-bit LCG s = s*a + c, stepped once per element before use (vectorised by block jumps)."""
    B = 4096
    first = np.empty(min(B, n), dtype=np.uint64)
    cur = s0
    AB, CB = 1, 0            # composition of B steps: s -> AB*s + CB
    for i in range(len(first)):
        cur = (cur * a + c) & 0xFFFFFFFF
        first[i] = cur
    for _ in range(B):
        AB, CB = (AB * a) & 0xFFFFFFFF, (CB * a + c) & 0xFFFFFFFF
    nblk = -(-n // B)
    out = np.empty((nblk, len(first)), dtype=np.uint64)
    out[0] = first
    for k in range(1, nblk):
        out[k] = (out[k - 1] * np.uint64(AB) + np.uint64(CB)) & np.uint64(0xFFFFFFFF)
    return out.reshape(-1)[:n].astype(np.uint32)


def recipe_p1(N, B, dim, pb_shift=8):
    """P1: cubic image of side N, blocks of side B, LCG pb.  pb_shift=16 gives recipe P4's pb."""
    shape = (N,) * dim
    nb = -(-N // B)
    idx = np.indices(shape)  # numpy order: (z,y,x) / (y,x)
    coords = idx[::-1]       # x, y, z
    label = np.ones(shape, dtype=np.int64)
    for d in range(dim):
        label += (coords[d] // B) * nb ** d
    out = _lcg(N ** dim)
    if pb_shift == 8:
        pb = ((out >> 8).astype(np.float32) / np.float32(16777216.0)).astype(np.float32)
    else:
        pb = ((out >> 16).astype(np.float32) / np.float32(65536.0)).astype(np.float32)
    return label.astype(np.uint32), pb.reshape(shape)
