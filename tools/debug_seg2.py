import sys, os, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers
helpers.load_pkg()
from pn2_amd.mlp import chain_rows
from test_streaming import _mlp
rows_per_seg, cin, widths, pool_k = [1024] * 7, 256, [256, 256, 512], 1
convs, bns = _mlp(widths, cin, False, seed=5)
convs2, bns2 = copy.deepcopy(convs), copy.deepcopy(bns)
L = lambda cv, bn: [(c, b, True) for c, b in zip(cv, bn)]
rows = sum(rows_per_seg)
x = torch.randn(rows, cin, device="cuda", generator=torch.Generator("cuda").manual_seed(6))
seg_off = np.concatenate([[0], np.cumsum(rows_per_seg)]).tolist()
nseg = len(rows_per_seg)
def saved(y):
    t = [u for u in y.grad_fn.saved_tensors if u is not None]
    ys = [u for u in t if u.dim() == 2 and u.shape[0] == y.shape[0] and u.shape[1] in widths]
    st = [u for u in t if u.dim() == 2 and u.shape[0] % 8 == 0 and u.shape[0] <= 8 * nseg and u.shape[1] in widths]
    return ys, st
xa = x.clone().requires_grad_(True)
ya = chain_rows(xa, L(convs, bns), pool_k=pool_k, seg_off=seg_off)
gout = torch.randn(ya.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(7))
ya.backward(gout, retain_graph=True)
ysa, sta = saved(ya)
print("saved y", [tuple(u.shape) for u in ysa], "stats", [tuple(u.shape) for u in sta])
s = 3
sl = slice(seg_off[s], seg_off[s + 1])
xb = x[sl].clone().requires_grad_(True)
yb = chain_rows(xb, L(convs2, bns2), pool_k=pool_k)
yb.backward(gout[sl], retain_graph=True)
ysb, stb = saved(yb)
print("out diff", float((ya[sl] - yb).abs().max()))
for li in range(len(ysb)):
    print("layer", li, "y diff", float((ysa[li][sl] - ysb[li]).abs().max()))
for li, (a, b) in enumerate(zip(sta, stb)):
    a = a.view(nseg, 8, -1)[s]
    print(f"layer {li}:", [f"{float((a[r] - b[r]).abs().max()):.2e}/{float(b[r].abs().max()):.2e}" for r in range(7)])
# float64 recomputation of layer 1's backward sums in segment s: dz1 = dY2 @ W2
W = [c.weight.double().reshape(c.out_channels, -1) for c in convs2]
g2, be2 = bns2[2].weight.double(), bns2[2].bias.double()
y2 = ysb[3].double(); y1 = ysb[2].double()
m2, v2 = y2.mean(0), y2.var(0, unbiased=False); is2 = 1 / torch.sqrt(v2 + bns2[2].eps)
xh2 = (y2 - m2) * is2
t2 = xh2 * g2 + be2
dzh2 = gout[sl].double() * (t2 > 0)
s1_2, s2_2 = dzh2.sum(0), (dzh2 * xh2).sum(0)
n = y2.shape[0]
dy2 = g2 * is2 * (dzh2 - s1_2 / n - xh2 * s2_2 / n)
dz1 = dy2 @ W[2]
g1, be1 = bns2[1].weight.double(), bns2[1].bias.double()
m1, v1 = y1.mean(0), y1.var(0, unbiased=False); is1 = 1 / torch.sqrt(v1 + bns2[1].eps)
xh1 = (y1 - m1) * is1
dzh1 = dz1 * ((xh1 * g1 + be1) > 0)
A1 = dzh1.sum(0) / n; B1 = is1 * (dzh1 * xh1).sum(0) / n
fa = sta[1].view(nseg, 8, -1)[s]; fb = stb[1]
print("layer1 A: fused err", float((fa[5].double() - A1).abs().max()), "separate err", float((fb[5].double() - A1).abs().max()), "scale", float(A1.abs().max()))
print("layer1 B: fused err", float((fa[6].double() - B1).abs().max()), "separate err", float((fb[6].double() - B1).abs().max()), "scale", float(B1.abs().max()))
bad = (fa[5].double() - A1).abs()
print("bad channels (A):", torch.nonzero(bad > 1e-6).flatten().tolist()[:40], "count", int((bad > 1e-6).sum()))

badB = (fa[6].double() - B1).abs()
print("bad channels (B):", torch.nonzero(badB > 1e-6).flatten().tolist()[:40], "count", int((badB > 1e-6).sum()))
print("A fused/true ratio on bad:", (fa[5].double() / A1)[bad > 1e-6][:10].tolist())
err = (xa.grad[sl] - xb.grad).abs()
print("dx err per 64-row chunk:", ["%.1e" % float(err[r:r + 64].max()) for r in range(0, 1024, 64)])
# which chunk's contribution is off?  per-chunk s1 of channel 118 in float64
c = 118
per_chunk = dzh1[:, c].view(-1, 64).sum(1)
print("true per-chunk s1[118]:", ["%.4f" % float(v) for v in per_chunk])
print("true total", float(per_chunk.sum()), "fused total", float(fa[5][c]) * n, "diff", float(fa[5][c]) * n - float(per_chunk.sum()))
