"""BASELINE configs[3]: the PointTransformerV3 backbone (reference defaults: five encoder stages 32..512 wide with depths 2/2/2/6/2,
four decoder stages, patches of 1024, four serialization orders) forward at inference on a ~1 M-voxel plot, one MI355X.  One JSON
line: forward time (HIP events around the whole call), how much of it the library's own kernels take (serialization codes,
neighbour tables, submanifold convolutions, patch attention) and how much the plain torch layers, and the float64 restatement
timed beside it on a bounded sample.
    python tools/bench_ptv3_model.py [--points 1048576 --grid 0.02 --reps 5 --precision f32|bf16]"""
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


def plot_voxels(points, grid, trees=16):
    from pn2_amd.synthetic import gaussian_branch_tree
    per = points * 3 // trees
    pts = np.concatenate([gaussian_branch_tree(per, seed=s)[0] + np.array([6.0 * (s % 4), 6.0 * (s // 4), 0.0], np.float32)
                          for s in range(trees)])
    g = np.unique(np.floor((pts - pts.min(0)) / grid).astype(np.int64), axis=0)
    rng = np.random.default_rng(0)
    return g[rng.permutation(len(g))[:points]]


def train_line(args, model, data, N, _hip):
    """One training step of the backbone: train-mode forward (batch-statistics BatchNorm, DropPath 0.3 spread over the blocks, the
    poolings' random choice of serialization), loss = mean square of the output rows, backward through every stage (attention
    and submanifold-conv backward kernels of this library, torch autograd for the dense layers), fused AdamW."""
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(data()).feat
        out.square().mean().backward()
        opt.step()
        return out
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
    ev[0].record()
    for i in range(args.reps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps))[args.reps // 2]
    groups = _hip.kernel_profile(lambda: (step(), torch.cuda.synchronize()))
    lib = {}
    for r in groups:
        lib[r["name"]] = lib.get(r["name"], 0.0) + r["ms"]
    print(json.dumps({
        "metric": "voxels/sec, PointTransformerV3 backbone fwd+bwd+AdamW (training)", "value": N / (ms * 1e-3), "unit": "voxels/s", "n_gpus": 1,
        "ms_per_step": ms, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3] as a training step: PTv3 backbone (reference defaults, in_channels 4, drop_path 0.3), {N} "
                               f"voxels of a 16-tree plot on a {args.grid} m grid, one cloud; output [N, {out.shape[1]}]"},
        "library_kernels_ms": {k: round(v, 3) for k, v in sorted(lib.items(), key=lambda kv: -kv[1])},
        "library_kernels_ms_total": round(sum(lib.values()), 3),
        "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
        "note": "the rest of the step is plain torch: Linear / LayerNorm / BatchNorm / GELU layers and their autograd, gathers, the "
                "pooling's unique / sort / segment reductions, the optimizer",
        "parity": "unpinned (reference module not importable here); tests/test_ptv3_train.py compares every parameter gradient with "
                  "the float64 restatement"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1 << 20)
    ap.add_argument("--grid", type=float, default=0.02)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--train", action="store_true", help="forward + backward + AdamW step in train mode instead of the inference forward")
    args = ap.parse_args()
    load_pkg()
    from pn2_amd import _hip
    from pn2_amd.PointTransformerV3 import attention as A
    from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerV3
    from pn2_amd.PointTransformerV3 import cpe
    A.ATTENTION_PRECISION = args.precision        # bf16: bfloat16 MFMA operands in the patch attention and in the submanifold
    cpe.CONV_PRECISION = args.precision           # convolutions of the layers >= 64 wide, and in the dense layers' GEMMs (fp32 rows)
    from pn2_amd import mlp
    mlp.GEMM_PRECISION = args.precision
    torch.manual_seed(0)
    model = PointTransformerV3(in_channels=4).cuda().eval()
    g = plot_voxels(args.points, args.grid)
    N = len(g)
    data = lambda: {"feat": feat, "coord": coord, "grid_coord": grid, "batch": batch}
    grid = torch.from_numpy(g.astype(np.int32)).cuda()
    coord = (grid.float() * args.grid)
    feat = torch.randn(N, 4, device="cuda")
    batch = torch.zeros(N, dtype=torch.int64, device="cuda")
    if args.train:
        return train_line(args, model, data, N, _hip)
    with torch.no_grad():
        for _ in range(2):
            out = model(data())
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
        ev[0].record()
        for i in range(args.reps):
            model(data())
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps))[args.reps // 2]
        groups = _hip.kernel_profile(lambda: model(data()) and torch.cuda.synchronize())
    lib = {}
    for r in groups:
        lib[r["name"]] = lib.get(r["name"], 0.0) + r["ms"]
    stages = [int(v) for v in []]
    # float64 restatement on a bounded sample of the same plot (the same model, a few thousand voxels)
    import bench
    from oracle import ptv3_model_port as P
    torch.set_num_threads(bench.host_cores())
    n_cpu = args.cpu_sample
    cfg = dict(order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2, 2, 2), enc_depths=(2, 2, 2, 6, 2), enc_num_head=(2, 4, 8, 16, 32),
               enc_patch_size=(1024,) * 5, dec_depths=(2, 2, 2, 2), dec_num_head=(4, 4, 8, 16), dec_patch_size=(1024,) * 4)
    gs = g[np.argsort(np.abs(g - g.mean(0)).sum(1))[:n_cpu]]          # a compact neighbourhood, so that voxels have neighbours
    t0 = time.perf_counter()
    P.backbone_forward(model.state_dict(), cfg, np.ones((n_cpu, 4), np.float32), gs * args.grid, gs, np.zeros(n_cpu, np.int64))
    dt = time.perf_counter() - t0
    out_line = {"metric": "voxels/sec, PointTransformerV3 backbone forward (inference)", "value": N / (ms * 1e-3), "unit": "voxels/s",
                "n_gpus": 1, "ms_per_step": ms, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": f"BASELINE configs[3]: PTv3 backbone (reference defaults, in_channels 4), {N} voxels of a 16-tree plot on a "
                                       f"{args.grid} m grid, one cloud, inference; output [N, {out.feat.shape[1]}]"},
                "library_kernels_ms": {k: round(v, 3) for k, v in sorted(lib.items(), key=lambda kv: -kv[1])},
                "library_kernels_ms_total": round(sum(lib.values()), 3),
                "note": "the rest of the forward is plain torch: Linear / LayerNorm / BatchNorm / GELU layers, gathers, the pooling's "
                        "unique / sort / segment reductions",
                "cpu_baseline": {"value": n_cpu / dt, "unit": "voxels/s", "cores": torch.get_num_threads(), "kind": "port",
                                 "sample": f"the {n_cpu} voxels nearest the plot's centre through the float64 restatement "
                                           f"(oracle/ptv3_model_port.py), {dt:.1f} s"},
                "parity": "unpinned (reference module not importable here); tests/test_ptv3_model.py compares with the restatement"}
    print(json.dumps(out_line))


if __name__ == "__main__":
    main()
