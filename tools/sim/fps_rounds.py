"""CPU study for the FPS design: rounds needed under the accepted-prefix rule for several caps K, with and without
per-member candidate limits, and the fraction of 1024-point Morton chunks a new centroid can touch (bounding-box test)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd.synthetic import gaussian_branch_tree

N, S = 262144, 1024
xyz, _, _ = gaussian_branch_tree(N, seed=0)
xyz = xyz.astype(np.float32)

def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / (hi - lo).max() * ((1 << bits) - 1)).astype(np.uint64)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(bits):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))

order = np.argsort(morton(xyz), kind="stable")
P = xyz[order]
CH = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nch = N // CH
chunk_lo = P.reshape(nch, CH, 3).min(1); chunk_hi = P.reshape(nch, CH, 3).max(1)
MEM = 32  # members: contiguous 8192-point spans
d = np.full(N, 1e10, np.float32)
cur = [0]
count = 1
stats = {K: 0 for K in (1, 2, 4, 8, 16)}
touched_hist = []
# sequential exact FPS; at each state compute the accepted-prefix length (unbounded) to derive rounds per K greedily
# do it properly: simulate per K separately would be expensive; instead record, for each sample index, the "chain length available"
# starting there -> rounds(K) by greedy jumps.
samples = [0]
avail = []   # avail[i]: max number of next samples obtainable in one round after applying samples[:i+1]
c = P[0]
for i in range(S - 1):
    dist = ((P - c) ** 2).sum(1)
    # chunk touch test before update
    cmax = d.reshape(nch, CH).max(1)
    dd = np.maximum(np.maximum(chunk_lo - c, c - chunk_hi), 0)
    touched = ((dd ** 2).sum(1) < cmax).sum()
    touched_hist.append(touched)
    d = np.minimum(d, dist)
    # top 24 candidates
    top = np.argpartition(-d, 24)[:24]
    top = top[np.argsort(-d[top], kind="stable")]
    acc = 1
    for t in range(1, 17):
        ok = True
        for a in range(t):
            if ((P[top[t]] - P[top[a]]) ** 2).sum() < d[top[t]]:
                ok = False; break
        if not ok: break
        acc = t + 1
    avail.append(acc)
    nxt = top[0]
    samples.append(nxt)
    c = P[nxt]
avail = np.array(avail)
for K in (1, 2, 3, 4, 6, 8, 12, 16):
    i = 0; rounds = 0
    while i < S - 1:
        i += min(avail[i], K); rounds += 1
    print(f"K={K:2d}: rounds {rounds}  picks/round {(S-1)/rounds:.2f}")
th = np.array(touched_hist)
print("chunks", nch, "touched per centroid: mean", th.mean(), "median", np.median(th), "after first 64: mean", th[64:].mean(), "max", th[64:].max())
print("touched quantiles (after 64):", np.quantile(th[64:], [0.5, 0.9, 0.99]))
