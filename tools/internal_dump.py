"""Replays a dump of the merge loop's state (GLIA_HMT_DUMP_DIR, written when a loop stops with the internal error).

  python tools/internal_dump.py gpurun_out/dumps/internal_<pid>_<n>.bin

Prints the first merge of the order that joins a region which no longer exists, the edge record that still claims to be
in the queue for it, and the contraction at which that edge should have died.
"""
import sys
import numpy as np

EDGE = np.dtype([("u", "<u4"), ("v", "<u4"), ("posu", "<u4"), ("posv", "<u4"), ("mean", "<f8"), ("n", "<i4"), ("next", "<u4"),
                 ("hu", "<u4", 2), ("hv", "<u4", 2), ("sal", "<f8"), ("seq", "<u8")])
FAT = np.dtype([("eid", "<u4"), ("rs", "<u4"), ("n", "<u4"), ("pos", "<u4"), ("off", "<u4"), ("len", "<u4"), ("mean", "<f8")])
assert EDGE.itemsize == 64 and FAT.itemsize == 32


def main(path):
    raw = open(path, "rb").read()
    R, E0, nk, ne, npool, cond_n, werr, wn = (int(x) for x in np.frombuffer(raw, "<u8", 8))
    o = 64
    order = np.frombuffer(raw, "<u4", 3 * nk, o).reshape(nk, 3); o += 12 * nk
    er = np.frombuffer(raw, EDGE, ne, o); o += 64 * ne
    pool = np.frombuffer(raw, FAT, npool, o)
    print("regions %d, initial edges %d, merges written %d, edge records %d, list entries %d, condition sizes %d, window overflow %d"
          % (R, E0, nk, ne, npool, cond_n, werr))
    alive = np.zeros(2 * R + 32, bool); alive[:R] = True
    died_at = {}
    for k, (a, b, c) in enumerate(order):
        bad = [int(x) for x in (a, b) if x >= len(alive) or not alive[x]]
        if bad or c != R + k:
            print("merge %d = (%d, %d) -> %d: region(s) %s no longer exist (gone at merge %s)" % (k, a, b, c, bad, [died_at.get(x) for x in bad]))
            cand = np.nonzero(((er["u"] == min(a, b)) & (er["v"] == max(a, b))))[0]
            for e in cand:
                r = er[e]
                print("  record %d: u %d v %d mean %.17g n %d sal %.17g seq %#x (merge %d, cat %d, rs %d) next %#x" %
                      (e, r["u"], r["v"], r["mean"], r["n"], r["sal"], int(r["seq"]), int(r["seq"]) >> 32, (int(r["seq"]) >> 30) & 3, int(r["seq"]) & 0x3FFFFFFF, r["next"]))
            break
        alive[a] = alive[b] = False; alive[c] = True
        died_at[int(a)] = died_at[int(b)] = k
    else:
        print("every merge of the order joins two existing regions")
    live = er["seq"] != 0
    stale = live & ~(alive[np.minimum(er["u"], len(alive) - 1)] & alive[np.minimum(er["v"], len(alive) - 1)])
    print("records still in the queue: %d, of them with a region that is gone (by the order up to the stop): %d" % (int(live.sum()), int(stale.sum())))


if __name__ == "__main__":
    main(sys.argv[1])
