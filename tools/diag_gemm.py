"""Diagnostic (GPU): where does a 128-tile GEMM workgroup spend its time?  Needs tools/build_diag_gemm.sh (a -DPN2_GEMM_DIAG
build of mlp.hip: per-workgroup cycle stamps).  Prints, for the forward, dgrad and wgrad launch of one 262144 x 128 x 128 layer:
shader clock, launch span, per-workgroup medians of prologue / K loop / epilogue, and how many workgroups ran at a time."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import PKG_DIR  # noqa: E402
os.environ["PN2_LIB"] = os.path.join(PKG_DIR, "build_diag", os.environ.get("PN2_DIAG_LIB", "libpn2hip_gemm_diag.so"))
from __graft_entry__ import load_pkg  # noqa: E402
import torch, torch.nn as nn  # noqa: E402
load_pkg()
from pn2_amd import _hip, mlp  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
mlp.FUSED_GRAD_ACCUMULATION = False
lib = _hip.lib()
tab = np.zeros((16384, 8), dtype=np.uint64)


def read(clear=True):
    lib.pn2_gemm_diag_read(tab.ctypes.data_as(ctypes.c_void_p), int(clear))
    return tab.copy()


def report(name, t):
    t = t[t[:, 1] > 0].astype(np.float64)
    if not len(t):
        print(name, "no stamps")
        return
    wall0, wall1 = t[:, 0].min(), t[:, 5].max()
    span_us = (wall1 - wall0) / 100.0
    cyc = (t[:, 4] - t[:, 1])
    wall = (t[:, 5] - t[:, 0]) / 100.0
    mhz = np.median(cyc / np.maximum(wall, 1e-3))
    med = lambda a: float(np.median(a))
    # concurrency: workgroups alive at the midpoint of the launch
    mid = 0.5 * (wall0 + wall1)
    alive = int(((t[:, 0] <= mid) & (t[:, 5] >= mid)).sum())
    starts = np.sort((t[:, 0] - wall0) / 100.0)
    if os.environ.get("PN2_DIAG_MAP"):
        raw = tab[:768, 6].astype(np.uint64)
        hw, xcc = raw & np.uint64(0xFFFFFFFF), (raw >> np.uint64(32)) & np.uint64(0xF)
        key = [(int(x), int((h >> 13) & 7), int((h >> 12) & 1), int((h >> 8) & 15)) for h, x in zip(hw, xcc)]
        by = {}
        for i, k in enumerate(key):
            by.setdefault(k, []).append(i)
        print("   compute units used by the first 768 workgroups:", len(by), " e.g.", list(by.items())[:6])
    print(f"{name}: {len(t)} workgroups, span {span_us:.1f} us, shader clock ~{mhz:.0f} MHz, alive at mid-launch {alive}")
    lives = np.sort(wall)
    occ = wall.sum() / (span_us * 256.0)
    print(f"   life 10/50/90/100 %: {lives[len(lives) // 10]:.1f}/{lives[len(lives) // 2]:.1f}/{lives[9 * len(lives) // 10]:.1f}/{lives[-1]:.1f} us;"
          f" time-averaged workgroups per compute unit {occ:.2f}")
    print(f"   per workgroup (cycles, median): prologue {med(t[:, 2] - t[:, 1]):.0f} | K loop {med(t[:, 3] - t[:, 2]):.0f} | "
          f"epilogue {med(t[:, 4] - t[:, 3]):.0f} | life {med(cyc):.0f} = {med(wall):.1f} us;  last start at +{starts[-1]:.1f} us, "
          f"10/50/90 % of starts by +{starts[len(starts) // 10]:.1f}/{starts[len(starts) // 2]:.1f}/{starts[9 * len(starts) // 10]:.1f} us")


torch.manual_seed(0)
conv, bn = nn.Conv1d(128, 128, 1).cuda(), nn.BatchNorm1d(128).cuda().train()
layers = [(conv, bn, True)]
g = torch.randn(rows, 128, device="cuda")
for precision in os.environ.get("PN2_DIAG_PRECISIONS", "f32,bf16").split(","):
    mlp.GEMM_PRECISION = precision
    os.environ["PN2_BF16_STORAGE"] = "0"
    for need_dx in (True, False):
        x = torch.randn(rows, 128, device="cuda", requires_grad=need_dx)
        if not need_dx:
            conv.weight.requires_grad_(True)
        for _ in range(3):
            y = mlp.chain_rows(x, layers)
            y.backward(g)
        torch.cuda.synchronize()
        read()
        y = mlp.chain_rows(x, layers)
        torch.cuda.synchronize()
        fwd = read()
        y.backward(g)
        torch.cuda.synchronize()
        bwd = read()
        if need_dx:
            report(f"[{precision}] forward", fwd)
            report(f"[{precision}] dgrad  ", bwd)
        else:
            report(f"[{precision}] wgrad  ", bwd)
