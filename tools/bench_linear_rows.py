"""GPU: the dense layers of the PTv3 mirror by shape (rows, C_in -> C_out): library GEMMs vs the HBM floor vs torch's linear."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import _hip
from pn2_amd.PointTransformerV3.linear import Linear
for rows, cin, cout in ((1 << 20, 32, 96), (1 << 20, 32, 32), (1 << 20, 32, 128), (1 << 20, 128, 32), (930000, 64, 192), (930000, 64, 256),
                        (930000, 256, 64), (565000, 128, 384), (565000, 512, 128), (183000, 256, 768), (183000, 1024, 256)):
    lin = Linear(cin, cout).cuda()
    x = torch.randn(rows, cin, device="cuda", requires_grad=True)
    y = lin(x)
    go = torch.randn_like(y)
    y.backward(go, retain_graph=True)
    def run():
        yy = lin(x)
        yy.backward(go)
        torch.cuda.synchronize()
    t = {r["name"]: r["ms"] for r in _hip.kernel_profile(run)}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.no_grad():
        torch.nn.functional.linear(x, lin.weight, lin.bias)
        ev[0].record()
        for _ in range(5):
            torch.nn.functional.linear(x, lin.weight, lin.bias)
        ev[1].record()
    torch.cuda.synchronize()
    floor = 4.0 * rows * (cin + cout) / 8e12 * 1e3
    print(f"{rows:8d} x {cin:4d} -> {cout:4d}: fwd {t.get('gemm_fwd', 0):6.3f} ms (HBM floor {floor:5.3f}, torch {ev[0].elapsed_time(ev[1]) / 5:6.3f}), "
          f"dgrad {t.get('gemm_dgrad', 0):6.3f}, wgrad {t.get('gemm_wgrad', 0):6.3f}, colsum {t.get('colsum', 0):6.3f}")
