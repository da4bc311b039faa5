"""Time the MLP chains of the deep levels (depth-4 table, one 262144-point tree) one by one: cooperative launch vs the
launch-per-layer path, forward and backward, sum of the library's HIP-event brackets per call (kernel time, host excluded).   python tools/bench_chain.py [--reps 200]"""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

SHAPES = [("sa1", 32768, 7, [32, 32, 64], 32, 0), ("sa2", 8192, 67, [64, 64, 128], 32, 3), ("sa3", 2048, 131, [128, 128, 256], 32, 3),
          ("sa4", 512, 259, [256, 256, 512], 32, 3), ("fp4", 64, 768, [256, 256], 1, 0), ("fp3", 256, 384, [256, 256], 1, 0),
          ("fp2", 1024, 320, [256, 128], 1, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    load_pkg()
    from pn2_amd import _hip, mlp
    for name, rows, cin, widths, K, skip in SHAPES:
        if args.only and name not in args.only.split(","):
            continue
        torch.manual_seed(0)
        layers, c = [], cin
        for w in widths:
            layers.append((nn.Conv2d(c, w, 1).cuda(), nn.BatchNorm2d(w).cuda().train(), True))
            c = w
        x = torch.randn(rows, cin, device="cuda", requires_grad=True)
        line = f"{name:4s} rows {rows:6d} {cin}->{widths} K={K}:"
        for mode in ("coop", "plain"):
            if mode == "plain":
                os.environ["PN2_NO_COOP"] = "1"
            else:
                os.environ.pop("PN2_NO_COOP", None)
            y = mlp.chain_rows(x, layers, pool_k=K, dx_first_col=skip)
            g = torch.randn_like(y)
            for _ in range(5):
                mlp.chain_rows(x, layers, pool_k=K, dx_first_col=skip).backward(g)
            torch.cuda.synchronize()
            acc = {"f": 0.0, "b": 0.0, "nf": 0, "nb": 0}

            def fwd():
                acc["y"] = [mlp.chain_rows(x, layers, pool_k=K, dx_first_col=skip) for _ in range(args.reps)]
                torch.cuda.synchronize()

            def bwd():
                for yy in acc["y"]:
                    yy.backward(g)
                torch.cuda.synchronize()

            gf = _hip.kernel_profile(fwd)          # HIP-event brackets around every library launch
            gb = _hip.kernel_profile(bwd)
            tf, nf = sum(r["ms"] for r in gf), sum(r["calls"] for r in gf)
            tb, nb = sum(r["ms"] for r in gb), sum(r["calls"] for r in gb)
            line += f"  {mode}: fwd {1e3 * tf / args.reps:6.1f} us ({nf / args.reps:.0f} launches)  bwd {1e3 * tb / args.reps:6.1f} us ({nb / args.reps:.0f}) |"
        print(line, flush=True)


if __name__ == "__main__":
    main()
