"""GPU: Point.serialization's codes (first stage) for 1 048 576 voxels, four orders -- csrc/serialize.hip against the
numpy port of the reference's algorithm timed on a bounded sample beside it.
    python tools/bench_serialization.py > profiles/rNN_bench_serialization.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

load_pkg()
from oracle import serialization_port as S  # noqa: E402
from pn2_amd.PointTransformerV3 import serialization as ser  # noqa: E402
from pn2_amd.PointTransformerV3.serialization.default import _encode_many as enc_many  # noqa: E402


def main():
    n, depth, orders = 1 << 20, 16, list(S.ORDERS)
    rng = np.random.default_rng(0)
    grid = rng.integers(0, 1 << depth, size=(n, 3)).astype(np.int32)
    batch = np.zeros(n, np.int64)
    tg, tb = torch.as_tensor(grid, device="cuda"), torch.as_tensor(batch, device="cuda")
    for _ in range(3):
        ser.serialize(tg, tb, depth, orders)
    torch.cuda.synchronize()

    def timed(fn, reps=20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3

    t_codes = timed(lambda: enc_many(tg, tb, depth, orders))
    t_all = timed(lambda: ser.serialize(tg, tb, depth, orders))
    m = 1 << 16
    t0 = time.perf_counter()
    want = [S.encode(grid[:m], batch[:m], depth, o) for o in orders]
    t_cpu = time.perf_counter() - t0
    got = enc_many(tg[:m], tb[:m], depth, orders).cpu().numpy()
    algo_bytes = n * (12 + 8 + 8 * len(orders))
    print(json.dumps({
        "metric": "voxels/s, serialization codes (z, z-trans, hilbert, hilbert-trans; depth 16)", "n_points": n,
        "codes_s": t_codes, "voxels_per_s_codes": n / t_codes, "hbm_GBs_codes": algo_bytes / t_codes / 1e9,
        "hbm_frac_codes": algo_bytes / t_codes / 8e12,
        "codes_sort_inverse_s": t_all, "voxels_per_s_with_sort": n / t_all,
        "cpu_baseline": {"value": m / t_cpu, "unit": "voxels/s", "cores": 1, "kind": "port",
                         "sample": f"first {m} voxels, four orders, numpy port of the reference's algorithm, {t_cpu:.2f} s"},
        "gpu_over_cpu_codes": (n / t_codes) / (m / t_cpu), "codes_equal_to_cpu_port_on_sample": bool(all(np.array_equal(a, b) for a, b in zip(got, want))),
        "dtype": "int64", "data": "synthetic"}))


main()
