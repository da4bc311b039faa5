"""Diagnostic (GPU): per-phase cycle shares of one FPS step, from a -DPN2_FPS_DIAG build of fps.hip."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg, PKG_DIR
load_pkg()
from pn2_amd.synthetic import gaussian_branch_tree
lib = ctypes.CDLL(os.path.join(PKG_DIR, "build_diag", "libpn2hip_diag.so"))
lib.pn2_fps_workspace_bytes.restype = ctypes.c_size_t
vp, i64 = ctypes.c_void_p, ctypes.c_int64
lib.pn2_fps_f32.argtypes = [vp, i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_size_t, vp, vp]
for (N, npoint, B) in [(262144, 1024, 1), (65536, 1024, 8), (1024, 256, 1)]:
    xyz = np.stack([gaussian_branch_tree(N, seed=s)[0] for s in range(B)])
    x = torch.from_numpy(xyz.transpose(0, 2, 1).copy()).cuda()        # [B,3,N]
    start = torch.zeros(B, dtype=torch.long, device="cuda")
    idx = torch.empty(B, npoint, dtype=torch.int32, device="cuda")
    nb = lib.pn2_fps_workspace_bytes(B, N, npoint)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        st = lib.pn2_fps_f32(x.data_ptr(), x.stride(0), x.stride(2), x.stride(1), B, N, npoint, start.data_ptr(), idx.data_ptr(), None,
                             ws.data_ptr(), nb, None, torch.cuda.current_stream().cuda_stream)
        assert st == 0
        torch.cuda.synchronize()
    xcd = 8192 < N <= 524288
    d = (ws[64:64 + 144] if xcd else ws[16:16 + 64]).view(torch.int64).cpu().numpy()
    if xcd:
        print("  (xcd kernel, local =", int(d[8]), ", rounds =", int(d[9]), ", rounds in which wavefront 0 of member 0 was touched =", int(d[10]), ", cycles per touched round: box+update", int(d[11]) // max(int(d[10]), 1), "select", int(d[12]) // max(int(d[10]), 1), "; tail per round: rank count", int(d[13]) // int(d[9]), "bound+ranks+list", int(d[14]) // int(d[9]), "barrier2", int(d[15]) // int(d[9]), "list read", int(d[16]) // int(d[9]), "chain", int(d[17]) // int(d[9]), ")")
        npoint = max(int(d[9]), 1)          # per ROUND figures for the multi-pick kernel
    tot, rt = d[6], d[7]
    print(f"N={N} npoint={npoint} B={B}: {tot / npoint:.0f} cycles/step, clock {tot / (rt / 100.0):.0f} MHz, "
          + " ".join(f"{n}={v / npoint:.0f}" for n, v in zip(["compute", "bar1", "scan+pub", "poll", "reduce", "bar2"], d[:6])))
