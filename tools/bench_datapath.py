"""GPU: the two data-path rows either side of the hot path, each with its CPU counterpart timed beside it.

  f-2  closest-cylinder projection of a 262 144-point tree onto a 2 000-cylinder QSM (Modules/Projection.py:117-144):
       one launch of pn2_cylinder_project_f32 vs the reference-structured torch-CPU batches of 1024 (oracle/projection_port.py)
       on a bounded sample; roofline: 90 flop per (point, cylinder) pair against the fp32 vector peak.
  f-1  rasterising the tree into 1 m boxes and building the padded mini-batch buffers (rasters.build_stream) vs the
       reference's loops (oracle/raster_port.py: one boolean mask over the cloud per box, host padding) on a bounded sample.
    python tools/bench_datapath.py > profiles/rNN_bench_datapath.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

load_pkg()
from pn2_amd import Projection, rasters  # noqa: E402
from pn2_amd.synthetic import gaussian_branch_tree  # noqa: E402
from oracle import projection_port, raster_port  # noqa: E402

N, M = 262144, 2000
xyz, off, _ = gaussian_branch_tree(N, seed=0)
rng = np.random.default_rng(0)
start = xyz[rng.integers(0, N, M)].astype(np.float32)
end = (start + rng.normal(size=(M, 3)) * 0.4).astype(np.float32)
radius = rng.uniform(0.02, 0.3, M).astype(np.float32)
ids = np.arange(M, dtype=np.int32)
dev = torch.device("cuda")


def gpu_time(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


# ---- f-2
cyl = {"startX": start[:, 0], "startY": start[:, 1], "startZ": start[:, 2], "endX": end[:, 0], "endY": end[:, 1], "endZ": end[:, 2],
       "radius": radius, "ID": ids}
s_t, r_t, l_t, u_t, i_t = Projection.cylinder_tensors(cyl, dev)
pts = torch.from_numpy(xyz).to(dev)
t_proj = gpu_time(lambda: Projection.cylinder_project(pts, s_t, u_t, l_t, r_t, i_t))
cloud64 = np.concatenate([xyz, np.zeros((N, 1), np.float32)], 1).astype(np.float64)
Projection.generate_offset_cloud_cuda_batched(cloud64, cyl, dev)
t0 = time.perf_counter()
lab = Projection.generate_offset_cloud_cuda_batched(cloud64, cyl, dev)
t_e2e = time.perf_counter() - t0
sample = 16384
torch.set_num_threads(min(16, os.cpu_count() or 1))
t0 = time.perf_counter()
ref = projection_port.generate_offset_cloud(xyz[:sample].astype(np.float64), start, end, radius, ids)
t_cpu = time.perf_counter() - t0
same_ids = float((ref[:, 6] == lab[:sample, 6]).mean())
proj = {"points": N, "cylinders": M, "kernel_s": t_proj, "points_per_s_kernel": N / t_proj, "end_to_end_s_host_arrays": t_e2e,
        "points_per_s_end_to_end": N / t_e2e, "TFLOPs_equivalent": 90.0 * N * M / t_proj / 1e12,
        "frac_of_fp32_vector_peak": 90.0 * N * M / t_proj / 1e12 / 157.3,
        "cpu_baseline": {"value": sample / t_cpu, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
                         "sample": f"first {sample} points against the same {M} cylinders, batches of 1024, torch CPU, {t_cpu:.2f} s"},
        "gpu_over_cpu": (N / t_e2e) / (sample / t_cpu), "ids_equal_to_cpu_port_on_sample": same_ids}

# ---- f-1
feats = torch.ones(N, 4, device=dev)
omask = torch.ones(N, dtype=torch.bool, device=dev)
for _ in range(3):                                                     # first calls load the sort / scan code objects
    rasters.build_stream(pts, feats, omask, 1.0, 1.0, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    stream = rasters.build_stream(pts, feats, omask, 1.0, 1.0, 10)
torch.cuda.synchronize()
t_rast = (time.perf_counter() - t0) / 20
sub = 32768
xs = xyz[:sub].astype(np.float64)
t0 = time.perf_counter()
b = raster_port.rasterize_clouds(xs, 1.0, 1.0)
rr = raster_port.getitem_rasters(xs.astype(np.float32), np.ones((sub, 4), np.float32), np.ones(sub, bool), b)
mb = raster_port.collate_streaming([r for r in rr if len(r["points"])], 10)
t_cpu_r = time.perf_counter() - t0
rast = {"points": N, "rasters": stream.flat["rasters"], "mini_batches": len(stream), "padded_points": int(sum(stream.flat["lengths"])),
        "build_stream_s": t_rast, "points_per_s": N / t_rast,
        "cpu_baseline": {"value": sub / t_cpu_r, "unit": "points/s", "cores": 1, "kind": "port",
                         "sample": f"the reference's loops (box masks over the cloud, host padding) on the first {sub} points "
                                   f"({len(b)} boxes, {len(mb)} mini-batches), numpy, {t_cpu_r:.2f} s"},
        "gpu_over_cpu": (N / t_rast) / (sub / t_cpu_r)}
# ---- BASELINE configs[4] on one GPU: predict_all_trees + kNN-to-QSM projection over a synthetic forest of 100 trees x ~8 000 points
# (two depth-5 models per tree through forward_hierarchical_streaming, 1 m rasters, mini-batches of 60, then the cylinder
# projection of the executed cloud): trees per second of ONE rank -- the sharded run hands each of the 8 ranks 12-13 of these trees.
from pn2_amd import predict  # noqa: E402
from pn2_amd.PointNet2.PointNet2 import PointNet2  # noqa: E402

n_trees = 100
trees, qsms = [], []
for t in range(n_trees):
    n_t = 7000 + 20 * t
    txyz, _, _ = gaussian_branch_tree(n_t, seed=100 + t)
    trees.append(txyz.astype(np.float64))
    r2 = np.random.default_rng(t)
    m = 200
    st_ = txyz[r2.integers(0, n_t, m)]
    en_ = st_ + r2.normal(size=(m, 3)) * 0.4
    qsms.append({"startX": st_[:, 0], "startY": st_[:, 1], "startZ": st_[:, 2], "endX": en_[:, 0], "endY": en_[:, 1], "endZ": en_[:, 2],
                 "radius": r2.uniform(0.02, 0.3, m), "ID": np.arange(m)})
torch.manual_seed(0)
m_off, m_noise = PointNet2(depth=5).cuda().eval(), PointNet2(depth=5).cuda().eval()
predict.predict_forest(m_off, m_noise, trees[:3], qsms[:3], seed=0)          # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
res = predict.predict_forest(m_off, m_noise, trees, qsms, seed=0)
torch.cuda.synchronize()
t_forest = time.perf_counter() - t0
forest = {"trees": n_trees, "points": int(sum(len(t) for t in trees)), "seconds": t_forest, "trees_per_s": n_trees / t_forest,
          "points_per_s": sum(len(t) for t in trees) / t_forest,
          "note": "one rank, host arrays in, numpy results out (BASELINE configs[4] shape; the 8-GPU run shards the trees, no collective)"}
print(json.dumps({"metric": "data path either side of the hot path (SURVEY 8 f-1 raster stream, f-2 cylinder projection), points/s",
                  "projection": proj, "raster_stream": rast, "forest_prediction_configs4": forest, "dtype": "f32", "data": "synthetic"}))
