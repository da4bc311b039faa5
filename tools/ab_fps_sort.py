"""A/B (GPU): multi-pick FPS in index order (PN2_FPS_NO_SORT=1) against the spatially ordered kernel, same process, HIP events.
Indices must be identical; PN2_FPS_KM selects the picks-per-round cap of the ordered kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import _hip
from pn2_amd.synthetic import gaussian_branch_tree
lib = _hip.lib()
cases = [(262144, 1024, 1), (65536, 1024, 8), (131072, 512, 2), (100000, 1024, 1), (262144, 4096, 1), (20000, 256, 3)]
for (N, npoint, B) in cases:
    xyz = np.stack([gaussian_branch_tree(N, seed=s)[0] for s in range(B)])
    x = torch.from_numpy(xyz.transpose(0, 2, 1).copy()).cuda()
    start = torch.arange(B, dtype=torch.long, device="cuda") * 7
    out = {}
    for tag, env in (("index", {"PN2_FPS_NO_SORT": "1"}), ("morton4", {"PN2_FPS_PER": "1"}), ("morton8", {})):
        for k in ("PN2_FPS_NO_SORT", "PN2_FPS_PER"):
            os.environ.pop(k, None)
        os.environ.update(env)
        idx = torch.empty(B, npoint, dtype=torch.int32, device="cuda")
        nxyz = torch.empty(B, npoint, 3, device="cuda")
        nb = lib.pn2_fps_workspace_bytes(B, N, npoint)
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.pn2_fps_f32(x.data_ptr(), x.stride(0), x.stride(2), x.stride(1), B, N, npoint, start.data_ptr(),
                                       idx.data_ptr(), nxyz.data_ptr(), ws.data_ptr(), nb, None, s)
        for _ in range(3):
            rc = call()
            assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        out[tag] = (e0.elapsed_time(e1) / 10 * 1e3, idx.cpu(), nxyz.cpu())
    same4 = torch.equal(out["index"][1], out["morton4"][1]) and torch.equal(out["index"][2], out["morton4"][2])
    same8 = torch.equal(out["index"][1], out["morton8"][1]) and torch.equal(out["index"][2], out["morton8"][2])
    print(f"N={N} npoint={npoint} B={B}: index order {out['index'][0]:.1f} us, cell order, 1 listed per member {out['morton4'][0]:.1f} us (same={same4}), "
          f"2 listed {out['morton8'][0]:.1f} us (same={same8})", flush=True)
