"""GPU: add_features (kNN normals / curvature / density, Modules/Features.py:178-229) on a synthetic tree --
points/s end to end (numpy in, numpy out), per-kernel times against the fp64 vector peak, and the CPU restatement of
the reference's loop timed on a bounded sample of the same cloud.
    python tools/bench_features.py [N] [cpu_sample]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

load_pkg()
from pn2_amd import Features as F, _hip  # noqa: E402
from pn2_amd.synthetic import gaussian_branch_tree  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
SAMPLE = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
FP64_PEAK_TFLOPS = 78.6
xyz, off, _ = gaussian_branch_tree(N, seed=0)
cloud = np.concatenate([xyz.astype(np.float64), off.astype(np.float64), np.zeros((N, 1))], axis=1)
F.add_features(cloud[:4096].copy())
torch.cuda.synchronize()
t0 = time.perf_counter()
out = F.add_features(cloud)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
groups = _hip.kernel_profile(lambda: (F.add_features(cloud), torch.cuda.synchronize()))
kernels = {g["name"]: {"ms": g["ms"] / g["calls"], "calls": g["calls"],
                       "TFLOPs_fp64": g["flops"] / (g["ms"] / g["calls"] * 1e-3) / 1e12 if g["flops"] else None} for g in groups}
knn = kernels.get("knn_radius", {})
from oracle.features_port import add_features_port  # noqa: E402
t0 = time.perf_counter()
ref = add_features_port(cloud, sample=SAMPLE)
dc = time.perf_counter() - t0
same_density = bool(np.array_equal(ref[:, 11], out[:SAMPLE, 11]))
cur_err = float(np.abs(ref[:, 10] - out[:SAMPLE, 10]).max())
print(json.dumps({
    "metric": "points/sec, add_features (k=15 normals, k=10 curvature, r=0.1 density, height, verticality, distance)",
    "value": N / dt, "unit": "points/s", "n_points": N, "seconds": dt, "dtype": "f64", "data": "synthetic",
    "kernels": kernels,
    "roofline": {"kernel": "knn_radius", "bound": "valu_fp64", "achieved": knn.get("TFLOPs_fp64"), "peak": FP64_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": (knn.get("TFLOPs_fp64") or 0) / FP64_PEAK_TFLOPS,
                 "work": "8 flop x N^2 distance tests (brute force)"},
    "cpu_baseline": {"value": SAMPLE / dc, "unit": "points/s", "cores": 1, "kind": "port",
                     "sample": f"first {SAMPLE} points of the same cloud (tree over all {N}), {dc:.1f} s"},
    "gpu_over_cpu": (N / dt) / (SAMPLE / dc),
    "check": {"density_equal_on_sample": same_density, "curvature_max_abs_err_on_sample": cur_err},
}))
