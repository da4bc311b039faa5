"""PointTransformerV3 conditional positional encoding (submanifold 3 x 3 x 3 sparse conv, reference blocks.py:561-568) at BASELINE
configs[3]'s size: ~1 M voxels of a synthetic Gaussian-branch forest plot on a 2 cm grid, rows in z-order (the order the serialized
backbone keeps them in), the encoder's stage widths.  One JSON line per width: neighbour-table time, conv time from the library's
HIP-event brackets, the useful matrix work (valid (voxel, offset) pairs only) against the fp32 MFMA peak, and the oracle's torch-CPU
restatement timed beside it on a bounded sample.
    python tools/bench_ptv3_cpe.py [--points 1048576 --grid 0.02 --widths 32,64,128 --reps 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_pkg  # noqa: E402

F32_MFMA_PEAK = 157.3      # TFLOP/s dense (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1 << 20)
    ap.add_argument("--grid", type=float, default=0.02)
    ap.add_argument("--widths", default="32,64,128")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    load_pkg()
    from pn2_amd import _hip
    from pn2_amd.PointTransformerV3 import cpe
    from pn2_amd.PointTransformerV3.serialization import encode
    from pn2_amd.synthetic import gaussian_branch_tree
    from oracle import ptv3_cpe_port as P
    # a plot of trees, voxelised; more raw points than voxels asked for (duplicates collapse)
    trees, per = 16, args.points * 3 // 16
    pts = np.concatenate([gaussian_branch_tree(per, seed=s)[0] + np.array([6.0 * (s % 4), 6.0 * (s // 4), 0.0], np.float32) for s in range(trees)])
    g = np.floor((pts - pts.min(0)) / args.grid).astype(np.int64)
    g = np.unique(g, axis=0)[: args.points]
    N = len(g)
    grid = torch.from_numpy(g.astype(np.int32)).cuda()
    batch = torch.zeros(N, dtype=torch.int64, device="cuda")
    depth = int(g.max()).bit_length()
    code = encode(grid, batch, depth, order="z")
    order = torch.argsort(code)
    grid, g = grid[order].contiguous(), g[order.cpu().numpy()]
    for _ in range(2):
        nbr = cpe.subm_neighbors(batch, grid)
    groups = _hip.kernel_profile(lambda: [cpe.subm_neighbors(batch, grid) for _ in range(args.reps)] and torch.cuda.synchronize())
    t_nbr = sum(r["ms"] for r in groups) / args.reps * 1e-3
    pairs = int((nbr >= 0).sum())
    cpu = None
    for C in [int(v) for v in args.widths.split(",")]:
        torch.manual_seed(0)
        conv = cpe.SubMConv3d(C, C).cuda()
        feat = torch.randn(N, C, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                conv(feat, nbr)
            groups = _hip.kernel_profile(lambda: [conv(feat, nbr) for _ in range(args.reps)] and torch.cuda.synchronize())
        gk = max(groups, key=lambda r: r["ms"])
        t = gk["ms"] / gk["calls"] * 1e-3
        flops = 2.0 * pairs * C * C
        if cpu is None:
            import bench
            torch.set_num_threads(bench.host_cores())
            sample = 65536
            t0 = time.perf_counter()
            nb = P.subm_neighbors(None, g[:sample])
            P.subm_conv(feat[:sample].cpu().numpy(), nb, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy())
            dt = time.perf_counter() - t0
            cpu = {"value": sample / dt, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"the first {sample} voxels (neighbour table + conv, C = {C}), numpy dict + torch float64 restatement, {dt:.2f} s"}
        out = {"metric": "voxels/sec, PTv3 CPE (submanifold 3x3x3 conv) forward", "value": N / t, "unit": "voxels/s", "n_gpus": 1,
               "ms_per_step": 1e3 * t, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{N} voxels of a 16-tree plot on a {args.grid} m grid in z-order, C = {C} -> {C}, "
                                      f"{pairs / N:.1f} of 27 neighbours present on average", "neighbour_table_ms": 1e3 * t_nbr},
               "roofline": {"kernel": gk["name"], "bound": "mfma", "achieved": flops / t / 1e12, "peak": F32_MFMA_PEAK, "unit": "TFLOP/s",
                            "frac": flops / t / 1e12 / F32_MFMA_PEAK, "traffic": None,
                            "note": "useful flops = 2 * valid (voxel, offset) pairs * C_in * C_out; a tile multiplies whole offset slabs, "
                                    "so the executed MFMA work is larger by the share of absent neighbours inside the slabs it keeps"},
               "cpu_baseline": cpu, "gpu_over_cpu": N / (t + t_nbr) / cpu["value"], "parity": "unpinned against spconv; pinned to torch conv3d"}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
