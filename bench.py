"""Headline benchmark: points/sec of PointNet2 forward + loss + backward on a synthetic 262 144-point tree.

    python bench.py [--gpus N --steps K --warmup W] [--mode monolithic|rasterized] [--depth D] [--points P] [--dtype f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` without a launcher starts the N ranks itself, as child processes, before touching the GPU.)

--mode monolithic (default; BASELINE configs[1]): a step = zero grads, PointNet2.forward on the whole tree as ONE cloud
  (depth 4: the training script's layer table), offset-regression loss, backward of 50*loss, one gradient all-reduce
  (N > 1) and the AdamW update.
--mode rasterized (what train_PointNet2.py --hierarchical --streaming does): the same tree cut
  into 1 m rasters, mini-batches of 10 rasters zero-padded to the group maximum, depth 5,
  PointNet2.forward_hierarchical_streaming with a backward per mini-batch (accumulated), ONE optimizer step per tree.

One pass of the hot path over one batch per step; one tree per rank (weak scaling, no data-path collective).  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line: the contract fields plus `roofline` (the dominant
libpn2hip kernel at its dominant launch shape, HIP-event timed on its launch stream during a second, instrumented run of the
same steps) and `cpu_baseline` (oracle/torch_port.py, the torch-CPU restatement of the reference path, timed on this
box's host cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
from __graft_entry__ import load_pkg  # noqa: E402

# library kernel name -> substring of the device kernel name in the PMC traffic summary (tools/pmc_traffic.py)
PMC_NAMES = {"fps": "fps_sorted_kernel", "gemm_fwd": "gemm_kernel<true, 1, true, 0, 0, 128", "gemm_dgrad": "gemm_kernel<true, 2, false, 0, 1, 128",
             "gemm_wgrad": "gemm_kernel<false, 2, false, 1, 2, 128", "three_interpolate_grad": "tig_reduce_kernel",
             "ball_query": "ball_query_kernel", "three_nn": "three_nn_kernel", "narrow_bwd": "narrow_bwd_kernel",
             "narrow_fwd": "narrow_fwd_kernel"}
PMC_FILES = [os.path.join(REPO, "profiles", f) for f in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")]


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source file) from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this command (separate passes, gfx950 corrections as in tools/pmc_traffic.py).  NOT measured in this run: PMC
    collection needs its own rocprofv3 passes; the source is named in the JSON line so a stale figure is visible."""
    key = PMC_NAMES.get(kernel)
    for path in PMC_FILES:
        try:
            table = json.load(open(path))
        except Exception:
            continue
        rows = [v for k, v in table.items() if key and key in k]
        if rows:
            return max(r["traffic_bytes_per_launch"] for r in rows), os.path.relpath(path, REPO)
    return None, None


HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
BF16_MFMA_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), the ceiling of the *_bf16 contractions
F32_MFMA_PEAK_TFLOPS = 157.3   # dense fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)
F32_VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak (same 64 FLOP/clk/SIMD rate): the ceiling of the distance-scan kernels
MFMA_KERNELS = ("gemm", "narrow")
METRIC = "points/sec fwd+bwd, PointNet2 offset-regression, 262k-pt tree, 1/2/4/8 GPUs"


def mfma_peak(kernel):
    return BF16_MFMA_PEAK_TFLOPS if kernel.endswith("_bf16") else F32_MFMA_PEAK_TFLOPS


def make_batch(n_points, seed, device, trees=1):
    """`trees` synthetic Gaussian-branch trees of n_points each (seeds seed*trees ... ), batched [B,3,N]."""
    import numpy as np
    from pn2_amd.synthetic import gaussian_branch_tree
    clouds = [gaussian_branch_tree(n_points, seed=seed * trees + t) for t in range(trees)]
    xyz = np.stack([c[0].T for c in clouds])                                       # [B,3,N] raw metres
    off = np.concatenate([c[1] for c in clouds])
    total = trees * n_points
    return {
        "coords": torch.from_numpy(xyz.copy()).to(device),
        "feats": torch.ones(trees, 4, n_points, device=device),                    # the reference's dummy features
        "masks_pad": torch.ones(trees, n_points, dtype=torch.bool, device=device),
        "masks_off": torch.ones(total, dtype=torch.bool, device=device),
        "semantic_labels": torch.zeros(total, dtype=torch.long, device=device),
        "offset_labels": torch.from_numpy(off).to(device),
    }


def make_raster_stream(n_points, seed, device, mbs=10):
    """The tree as the reference's streaming collate hands it over (RasterizedTreeSet.py:390-459): 1 m rasters
    (size = stride = 1.0), mini-batches of `mbs` rasters zero-padded to the group's longest, raw metres, dummy features
    (zero on padding).  -> (list of mini-batch dicts on `device`, label dict, padded point count)."""
    import numpy as np
    from pn2_amd.synthetic import gaussian_branch_tree, rasterize
    xyz, off, _ = gaussian_branch_tree(n_points, seed=seed)
    rasters = rasterize(xyz, 1.0, 1.0)
    out, padded = [], 0
    for k in range(0, len(rasters), mbs):
        group = rasters[k:k + mbs]
        nmax = max(len(r) for r in group)
        coords = np.zeros((len(group), 3, nmax), np.float32)
        mpad = np.zeros((len(group), nmax), bool)
        for i, r in enumerate(group):
            coords[i, :, :len(r)] = xyz[r].T
            mpad[i, :len(r)] = True
        ids = np.concatenate(group)
        pad_t = torch.from_numpy(mpad).to(device)
        out.append({"coords": torch.from_numpy(coords).to(device), "feats": torch.ones(len(group), 4, nmax, device=device) * pad_t[:, None, :],
                    "masks_pad": pad_t, "masks_off": torch.ones(len(ids), dtype=torch.bool, device=device),
                    "point_ids": torch.from_numpy(ids).to(device)})
        padded += len(group) * nmax
    labels = {"cloud_length": n_points, "semantic_labels": torch.zeros(n_points, 1, dtype=torch.long, device=device),
              "offset_labels": torch.from_numpy(off).to(device)}
    return out, labels, padded, len(rasters)


def host_cores():
    """CPU cores this process may actually use: the cgroup quota when there is one (a GPU box hands a job a share of
    the host -- 16 of 256 logical CPUs on the MI355X pool; 128 oversubscribed threads ran the baseline 2.3x SLOWER
    than 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(depth, n_points, seed, trees=1):
    """One full step of the same workload through the torch-CPU restatement of the reference path, on as many threads
    as the job has cores."""
    from oracle import torch_port as P
    torch.set_num_threads(host_cores())
    batch = {k: v.cpu() for k, v in make_batch(n_points, seed, "cpu", trees).items()}
    torch.manual_seed(0)
    model = P.PortPointNet2(depth=depth).train()
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    times = []
    for _ in range(2):                      # the second repetition runs warm; the better one is reported
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _, _, _ = P.loss_from_batch(model, batch, mult_sem=0.0)
        (loss * 50).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    dt = min(times)
    return {"value": trees * n_points / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"best of 2 steps (fwd+loss+bwd+AdamW) of the same depth-{depth} batch of {trees} x {n_points}-point "
                      f"tree(s), torch CPU fp32, {times[0]:.2f} s and {times[1]:.2f} s"}


def cpu_baseline_rasterized(depth, n_points, seed):
    """The reference's streaming loop (PointNet2.py:238-306) restated on the CPU port: per mini-batch forward, offset loss
    on the real points, backward of 50*loss (accumulated), one AdamW step per tree.  One whole tree = the bounded sample."""
    from oracle import torch_port as P
    torch.set_num_threads(host_cores())
    stream, labels, padded, n_rasters = make_raster_stream(n_points, seed, "cpu")
    torch.manual_seed(0)
    model = P.PortPointNet2(depth=depth).train()
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
    t0 = time.perf_counter()
    opt.zero_grad()
    for mb in stream:
        sem, off = model(mb["coords"], mb["feats"])
        keep = mb["masks_pad"].reshape(-1)
        off_v = off.permute(0, 2, 1).reshape(-1, 3)[keep][mb["masks_off"]]
        lab = labels["offset_labels"][mb["point_ids"]][mb["masks_off"]]
        _, lo = P.point_wise_loss(sem.permute(0, 2, 1).reshape(-1, 2)[keep], off_v, labels["semantic_labels"].squeeze()[mb["point_ids"]], lab)
        (lo * 50).backward()
    opt.step()
    dt = time.perf_counter() - t0
    return {"value": n_points / dt, "unit": "points/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"one whole tree in the reference's streaming loop: {len(stream)} mini-batches of 10 rasters "
                      f"({n_rasters} rasters, {padded} padded points), depth {depth}, fwd + offset loss + bwd per mini-batch, "
                      f"one AdamW step, torch CPU fp32, {dt:.2f} s"}


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` typed as such (no launcher): start N ranks as CHILD processes with torch.distributed.run --
    before this process has made any GPU call (a process that has touched the GPU must never be re-exec'ed, and this one
    is not: it only waits) -- and hand rank 0's JSON line through.  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["monolithic", "rasterized"], default="monolithic")
    ap.add_argument("--depth", type=int, default=None, help="reference layer table (default: 4 monolithic = train_PointNet2.py's, "
                                                            "5 rasterized = the shipping table)")
    ap.add_argument("--points", type=int, default=262144)
    ap.add_argument("--trees", type=int, default=1, help="monolithic: trees per GPU per step (BASELINE configs[2]: 8 x 65536)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = the reference's arithmetic (parity mode); bf16 = bfloat16 MFMA operands with fp32 accumulation in the "
                         "large contractions (throughput mode, BASELINE configs[1]; tolerance: tests/test_bf16_mode.py)")
    ap.add_argument("--graph", action="store_true",
                    help="monolithic, 1 GPU: the whole step (zero grads, forward, loss, backward, AdamW) captured once as ONE HIP graph "
                         "and replayed per step with fresh FPS start indices (pn2_amd/graphs.py) -- same kernels, no host launch path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-groups", default=None, help="write every (kernel, launch shape) group of the instrumented run here")
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, argv))
    rasterized = args.mode == "rasterized"
    depth = args.depth if args.depth is not None else (5 if rasterized else 4)

    load_pkg()
    from pn2_amd import _hip, mlp, ops, parallel
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    mlp.GEMM_PRECISION = args.dtype

    rank, local, world = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("PN2_DIST_BACKEND") == "gloo":      # rehearsal of the N > 1 path on a one-GPU box
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)                                  # identical random-init weights on every rank
    model = PointNet2(depth=depth, loss_multiplier_semantic=0).to(dev).train()
    # parameters and gradients live in two flat, 16-byte-aligned buffers: one all-reduce and ONE fused AdamW launch per step
    grads = parallel.FlatGradAllReduce(model, flatten_params=True)
    use_graph = args.graph and world == 1 and not rasterized
    opt = torch.optim.AdamW(grads.optimizer_params(), lr=0.01, weight_decay=1e-3, fused=True,   # train_PointNet2.py:250
                            capturable=use_graph)
    if rasterized:
        stream, labels, padded, n_rasters = make_raster_stream(args.points, seed=rank, device=dev)
    else:
        batch = make_batch(args.points, seed=rank, device=dev, trees=args.trees)
    torch.manual_seed(1000 + rank)                        # FPS start indices: per-rank stream

    class Scaler:                                         # fp32: the reference's GradScaler would scale by 1
        def scale(self, x):
            return x

    def step():
        grads.zero()
        if rasterized:
            loss, _ = model.forward_hierarchical_streaming(dict(labels, mini_batches=iter(stream)), return_loss=True, scaler=Scaler())
        else:
            loss, _ = model(batch, return_loss=True)
            (loss * 50).backward()                        # train_utils.py:57-58
        grads.allreduce()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step_eager = step
    if use_graph:
        from pn2_amd.graphs import GraphedTrainStep
        step = GraphedTrainStep(step_eager, warmup=2)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    ops.check_status()                                    # no kernel gave up during the timed steps
    assert (loss == loss) if isinstance(loss, float) else torch.isfinite(loss).item()

    # second, instrumented run of the same steps: the library brackets every kernel launch with HIP events on its
    # launch stream (pn2_prof_enable) and aggregates per (kernel, launch shape)
    def timed_steps():
        for _ in range(args.steps):
            step_eager()                                  # (eager: a graph replay makes no library calls to bracket)
        torch.cuda.synchronize()
    groups = _hip.kernel_profile(timed_steps)
    barrier()

    if rank == 0 and args.dump_groups:
        json.dump(sorted(groups, key=lambda g: -g["ms"]), open(args.dump_groups, "w"), indent=0)
    if rank == 0:
        step_ms = 1e3 * dt / args.steps
        kernels = {}
        for gr in groups:
            k = kernels.setdefault(gr["name"], {"launches_per_step": 0.0, "ms_per_step": 0.0})
            k["launches_per_step"] += gr["calls"] / args.steps
            k["ms_per_step"] += gr["ms"] / args.steps
        for k in kernels.values():
            k["share_of_step"] = k["ms_per_step"] / step_ms
        # every library kernel at its heaviest launch shape against both ceilings: "mfma" (fp32 matrix peak) for the
        # contractions, "valu" (fp32 vector peak) for the distance scans
        per_kernel = {}
        for gr in groups:
            cur = per_kernel.get(gr["name"])
            if cur is None or gr["ms"] > cur["ms"]:
                per_kernel[gr["name"]] = gr
        rooflines = {}
        for name, gr in per_kernel.items():
            t = gr["ms"] / gr["calls"] * 1e-3
            row = {"avg_launch_us": 1e6 * t, "launches_per_step": gr["calls"] / args.steps,
                   "hbm_GBs": gr["bytes"] / t / 1e9, "hbm_frac": gr["bytes"] / t / 1e9 / HBM_PEAK_GBS}
            if gr["flops"] > 0:
                mfma = name.startswith(MFMA_KERNELS)
                row["TFLOPs"] = gr["flops"] / t / 1e12
                row["compute_unit"] = ("mfma_bf16" if name.endswith("_bf16") else "mfma") if mfma else "valu"
                row["compute_frac"] = row["TFLOPs"] / (mfma_peak(name) if mfma else F32_VALU_PEAK_TFLOPS)
            rooflines[name] = row
        dom = max(groups, key=lambda gr: gr["ms"])            # dominant (kernel, shape) by total time
        avg_s = dom["ms"] / dom["calls"] * 1e-3
        hbm = dom["bytes"] / avg_s / 1e9
        tfl = dom["flops"] / avg_s / 1e12
        peak = mfma_peak(dom["name"])
        if dom["name"].startswith(MFMA_KERNELS) and tfl / peak > hbm / HBM_PEAK_GBS:
            roofline = {"kernel": dom["name"], "bound": "mfma", "achieved": tfl, "peak": peak, "unit": "TFLOP/s",
                        "frac": tfl / peak, "traffic": None, "hbm_GBs": hbm, "hbm_frac": hbm / HBM_PEAK_GBS}
        else:
            roofline = {"kernel": dom["name"], "bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm / HBM_PEAK_GBS, "traffic": None}
        if dom["name"] == "fps":
            # FPS is npoint strictly dependent argmax rounds over data that stays in registers: the ceiling is the
            # latency of a round, not HBM; the figure to drive down is microseconds per sample
            npoint = {4: 1024, 3: 1024, 2: 1024, 5: 100, 6: 500}.get(depth, 1024)
            roofline.update({"bound": "latency", "us_per_sample": 1e6 * avg_s / npoint, "samples": npoint,
                             "note": "hbm figures kept for the contract; the kernel reads its cloud once and is bound by dependent rounds"})
        traffic, source = pmc_traffic(dom["name"])
        if dom["name"] == "fps":
            roofline["rounds"] = ops.last_fps_rounds()     # exchanges of the level-1 call (up to 16 samples each)
        roofline["traffic"] = traffic
        roofline["traffic_source"] = (f"{source}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; "
                                      "not re-measured in this run") if source else None
        roofline.update({"algorithmic_bytes_per_launch": dom["bytes"], "algorithmic_flops_per_launch": dom["flops"],
                         "avg_launch_us": 1e6 * avg_s, "launches_per_step": dom["calls"] / args.steps,
                         "share_of_step": dom["ms"] / args.steps / step_ms})
        mode_text = ("fp32 parity mode" if args.dtype == "f32" else
                     "bf16-operand MFMA throughput mode (fp32 accumulation and storage; tolerance tests/test_bf16_mode.py)")
        if rasterized:
            workload = (f"PointNet2 depth {depth} fwd+loss+bwd+AdamW, the reference's streaming raster mode: one {args.points}-point "
                        f"Gaussian-branch tree per GPU as {n_rasters} one-metre rasters in {len(stream)} mini-batches of 10 "
                        f"({padded} padded points), backward per mini-batch, one optimizer step per tree (SURVEY 8d config 2(i)), "
                        f"{mode_text}")
        else:
            workload = (f"PointNet2 depth {depth} fwd+loss+bwd+AdamW, B={args.trees} x {args.points}-point Gaussian-branch tree(s) per "
                        f"GPU ({'BASELINE configs[1], monolithic' if args.trees == 1 else 'BASELINE configs[2] shape'}), {mode_text}")
        out = {
            "metric": METRIC,
            "value": args.points * args.trees * world * args.steps / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "mode": args.mode, "hip_graph": bool(use_graph), "points_per_gpu": args.points * args.trees,
                       "trees_per_gpu": args.trees, "depth": depth,
                       "parallelism": f"dp{world} ({args.trees} tree(s) per rank, 1 flat gradient all-reduce per step)"},
            "roofline": roofline, "kernels": kernels, "rooflines": rooflines,
            "library_launches_per_step": sum(k["launches_per_step"] for k in kernels.values()),
        }
        bq = rooflines.get("ball_query")
        out["north_star_ball_query"] = {
            "hbm_frac": bq["hbm_frac"] if bq else None, "avg_launch_us": bq["avg_launch_us"] if bq else None, "target": 0.7,
            "note": "north_star asks >= 0.7 of HBM on ball_query; its compulsory traffic B*(12N + 12S + 8SK) is 3.4 MB at 1 x 262144 "
                    "(68 MB at 64 x 65536) = 0.4 us at peak, less than a kernel launch: the op is search-latency bound and the "
                    "target is not meetable at these sizes (DESIGN.md section 3); the fraction is reported every run"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline_rasterized(depth, args.points, seed=0) if rasterized
                                   else cpu_baseline(depth, args.points, seed=0, trees=args.trees))
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
