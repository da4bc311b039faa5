"""A/B (GPU): round-1 FPS kernels (build_diag/libfps_r1.so) against the current library, HIP-event timed in one process."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg, PKG_DIR
load_pkg()
from pn2_amd import _hip
from pn2_amd.synthetic import gaussian_branch_tree
vp, i64 = ctypes.c_void_p, ctypes.c_int64
old = ctypes.CDLL(os.path.join(PKG_DIR, "build_diag", "libfps_r1.so"))
old.pn2_fps_workspace_bytes.restype = ctypes.c_size_t
old.pn2_fps_f32.argtypes = [vp, i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_size_t, vp]
new = _hip.lib()
for (N, npoint, B) in [(1024, 256, 1), (256, 64, 1), (64, 16, 1), (262144, 1024, 1), (65536, 1024, 8)]:
    xyz = np.stack([gaussian_branch_tree(N, seed=s)[0] for s in range(B)])
    x = torch.from_numpy(xyz.transpose(0, 2, 1).copy()).cuda()
    start = torch.zeros(B, dtype=torch.long, device="cuda")
    res = {}
    for tag, lib in (("r1", old), ("now", new), ("r1", old), ("now", new)):
        idx = torch.empty(B, npoint, dtype=torch.int32, device="cuda")
        nxyz = torch.empty(B, npoint, 3, device="cuda")
        nb = lib.pn2_fps_workspace_bytes(B, N, npoint)
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        def call():
            if tag == "r1":
                return lib.pn2_fps_f32(x.data_ptr(), x.stride(0), x.stride(2), x.stride(1), B, N, npoint, start.data_ptr(), idx.data_ptr(), nxyz.data_ptr(), ws.data_ptr(), nb, s)
            return lib.pn2_fps_f32(x.data_ptr(), x.stride(0), x.stride(2), x.stride(1), B, N, npoint, start.data_ptr(), idx.data_ptr(), nxyz.data_ptr(), ws.data_ptr(), nb, None, s)
        for _ in range(3):
            assert call() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(tag, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        res[tag + "_idx"] = idx.cpu()
    print(f"N={N} npoint={npoint} B={B}: r1 {min(res['r1']):.1f} us, now {min(res['now']):.1f} us, same={torch.equal(res['r1_idx'], res['now_idx'])}")
