import sys, os, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers
helpers.load_pkg()
from pn2_amd.mlp import chain_rows
from test_streaming import _mlp
def run(rows_per_seg, cin, widths, pool_k):
    convs, bns = _mlp(widths, cin, pool_k > 1, seed=5)
    convs2, bns2 = copy.deepcopy(convs), copy.deepcopy(bns)
    L = lambda cv, bn: [(c, b, True) for c, b in zip(cv, bn)]
    rows = sum(rows_per_seg)
    x = torch.randn(rows, cin, device="cuda", generator=torch.Generator("cuda").manual_seed(6))
    seg_off = np.concatenate([[0], np.cumsum(rows_per_seg)]).tolist()
    xa = x.clone().requires_grad_(True)
    ya = chain_rows(xa, L(convs, bns), pool_k=pool_k, seg_off=seg_off)
    gout = torch.randn(ya.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(7))
    ya.backward(gout)
    errs = []
    for s in range(len(rows_per_seg)):
        sl = slice(seg_off[s], seg_off[s + 1])
        xb = x[sl].clone().requires_grad_(True)
        yb = chain_rows(xb, L(copy.deepcopy(convs2), copy.deepcopy(bns2)), pool_k=pool_k)
        yb.backward(gout[seg_off[s] // pool_k:seg_off[s + 1] // pool_k])
        errs.append(float((xa.grad[sl] - xb.grad).abs().max()) / float(xb.grad.abs().max()))
    print(rows_per_seg, cin, widths, pool_k, ["%.1e" % e for e in errs])
run([2560, 2048, 2560], 256, [256, 256, 512], 32)
run([2560, 2048, 2560], 256, [256, 256, 512], 1)
run([2048, 2560], 256, [256, 256, 512], 1)
run([2560, 2560, 2560], 256, [256, 256, 512], 1)
run([1024, 1024, 1024, 1024, 1024, 1024, 1024], 256, [256, 256, 512], 1)
run([2560, 2048, 2560], 256, [256, 512], 1)
run([2560, 2048, 2560], 256, [256, 256, 256], 1)
run([2560, 2048, 2560], 128, [128, 128, 128], 1)
run([2560, 2048, 2560], 256, [512, 512], 1)
run([5120, 4096, 5120], 256, [256, 256, 512], 1)
print("--- PN2_WGRAD_BLOCKS=64")
os.environ["PN2_WGRAD_BLOCKS"] = "64"
run([2560, 2048, 2560], 256, [256, 256, 512], 1)
run([1024] * 7, 256, [256, 256, 512], 1)
print("--- PN2_WGRAD_BLOCKS=2048")
os.environ["PN2_WGRAD_BLOCKS"] = "2048"
run([2560, 2048, 2560], 256, [256, 256, 512], 1)
run([1024] * 7, 256, [256, 256, 512], 1)
