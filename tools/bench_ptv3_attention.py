"""PointTransformerV3 serialized patch attention at BASELINE configs[3]'s size (1 048 576 voxels, first encoder stage: C = 32,
2 heads of 16, patches of 1024; reference blocks.py:457-488): one JSON line per precision with the kernel's HIP-event time, its
roofline against the dense MFMA peak of the mode, and the torch-CPU restatement timed beside it on a bounded sample.
    python tools/bench_ptv3_attention.py [--points 1048576 --channels 32 --heads 2 --reps 20]"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from __graft_entry__ import load_pkg  # noqa: E402

F32_MFMA_PEAK, BF16_MFMA_PEAK = 157.3, 2500.0      # TFLOP/s dense (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1 << 20)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--patch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    load_pkg()
    from pn2_amd import _hip
    from pn2_amd.PointTransformerV3 import attention as A
    from oracle import ptv3_attention_port as P
    torch.manual_seed(0)
    N, C, H, K = args.points // args.patch * args.patch, args.channels, args.heads, args.patch
    qkv = torch.randn(N, 3 * C, device="cuda")
    order = torch.randperm(N, device="cuda")
    scale = (C // H) ** -0.5
    flops = 4.0 * N * K * C                                   # Q K^T and P V, once each (the algorithmic count)
    cpu = None
    for prec in ("f32", "bf16"):
        A.ATTENTION_PRECISION = prec
        for _ in range(3):
            A.patch_attention(qkv, order, K, H, scale)
        groups = _hip.kernel_profile(lambda: [A.patch_attention(qkv, order, K, H, scale) for _ in range(args.reps)] and torch.cuda.synchronize())
        g = max(groups, key=lambda r: r["ms"])
        t = g["ms"] / g["calls"] * 1e-3
        peak = BF16_MFMA_PEAK if prec == "bf16" else F32_MFMA_PEAK
        if cpu is None:
            sample = 32 * K                                    # 32 patches on the host cores
            import bench
            torch.set_num_threads(bench.host_cores())          # the job's share of the host (cgroup quota), not the whole box
            q = qkv[order[:sample]].cpu()
            t0 = time.perf_counter()
            P.patch_attention(q, None, K, H, scale)
            dt = time.perf_counter() - t0
            cpu = {"value": sample / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"{sample} points (32 patches of {K}), torch CPU fp32 restatement of blocks.py:466-481, {dt:.2f} s"}
        out = {"metric": "points/sec, PTv3 serialized patch attention forward", "value": N / t, "unit": "points/s", "n_gpus": 1,
               "ms_per_step": 1e3 * t, "dtype": prec, "data": "synthetic",
               "config": {"workload": f"{N} points, C = {C}, {H} heads of {C // H}, patches of {K}, gather by a random order fused"},
               "roofline": {"kernel": g["name"], "bound": "mfma", "achieved": flops / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                            "frac": flops / t / 1e12 / peak, "traffic": None,
                            "note": "the exact two-pass softmax computes Q K^T twice: executed MFMA work is 1.5x the algorithmic count"},
               "cpu_baseline": cpu, "gpu_over_cpu": N / t / cpu["value"], "parity": "unpinned (reference module not importable here)"}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
