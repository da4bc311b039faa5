"""CPU study: rounds needed when the round's accepted samples come from simulating sequential FPS on the LISTED candidates
(one best point per member, members = runs of sorted points) while the best updated listed key stays above H = the largest
member bound (second best of a member); optionally two listed candidates per member."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd.synthetic import gaussian_branch_tree

N, S, G = 262144, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 32
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # listed candidates per member
CAP = int(sys.argv[3]) if len(sys.argv) > 3 else 64
xyz, _, _ = gaussian_branch_tree(N, seed=0)
xyz = xyz.astype(np.float32)
lo, hi = xyz.min(0), xyz.max(0)
cell = np.minimum(((xyz - lo) / (hi - lo) * 16).astype(np.int64), 15)
def spread(v):
    out = np.zeros_like(v)
    for b in range(4):
        out |= ((v >> b) & 1) << (3 * b)
    return out
code = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
order = np.argsort(code, kind="stable")
P = xyz[order]
d = np.full(N, 1e10, np.float32)
d = np.minimum(d, ((P - P[0]) ** 2).sum(1))
count, rounds = 1, 0
span = N // G
hist = []
while count < S:
    rounds += 1
    dm = d.reshape(G, span)
    idx = np.argsort(-dm, axis=1)[:, :PER + 1]                    # top PER+1 per member
    cand = (idx[:, :PER] + np.arange(G)[:, None] * span).ravel()
    H = np.take_along_axis(dm, idx[:, PER:PER + 1], 1).max()
    cd = d[cand].copy()
    acc = []
    while len(acc) < min(CAP, S - count):
        j = int(np.argmax(cd))
        if acc and not cd[j] > H:
            break
        acc.append(cand[j])
        cd = np.minimum(cd, ((P[cand] - P[cand[j]]) ** 2).sum(1))
    for a in acc:
        d = np.minimum(d, ((P - P[a]) ** 2).sum(1))
    count += len(acc)
    hist.append(len(acc))
print(f"G={G} listed/member={PER} cap={CAP}: rounds {rounds}, picks/round {np.mean(hist):.2f}, max {max(hist)}")

# ---- variant: two listed per member, but taken from DIFFERENT wavefronts (each wavefront lists only its best point)
def variant_distinct_waves(NWV=8):
    d = np.full(N, 1e10, np.float32)
    d = np.minimum(d, ((P - P[0]) ** 2).sum(1))
    count, rounds, hist = 1, 0, []
    wspan = span // NWV
    while count < S:
        rounds += 1
        dw = d.reshape(G, NWV, wspan)
        ia = np.argsort(-dw, axis=2)[:, :, :2]
        best = np.take_along_axis(dw, ia[:, :, :1], 2)[:, :, 0]        # [G, NWV]
        second = np.take_along_axis(dw, ia[:, :, 1:2], 2)[:, :, 0]
        bidx = ia[:, :, 0] + np.arange(NWV)[None, :] * wspan + np.arange(G)[:, None] * span
        wo = np.argsort(-best, axis=1)
        cand = np.take_along_axis(bidx, wo[:, :2], 1).ravel()
        third = np.take_along_axis(best, wo[:, 2:3], 1)[:, 0]
        H = max(third.max(), second.max())
        cd = d[cand].copy()
        acc = []
        while len(acc) < min(16, S - count):
            j = int(np.argmax(cd))
            if acc and not cd[j] > H:
                break
            acc.append(cand[j])
            cd = np.minimum(cd, ((P[cand] - P[cand[j]]) ** 2).sum(1))
        for a in acc:
            d = np.minimum(d, ((P - P[a]) ** 2).sum(1))
        count += len(acc)
        hist.append(len(acc))
    print(f"distinct-wavefront variant: rounds {rounds}, picks/round {np.mean(hist):.2f}")

if G == 32 and PER == 2:
    variant_distinct_waves()
