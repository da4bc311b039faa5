"""GPU: which torch (non-library) kernels run in one headline step, with shapes and call sites (torch.profiler)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import parallel
from pn2_amd.PointNet2.PointNet2 import PointNet2
dev = torch.device("cuda")
torch.manual_seed(0)
model = PointNet2(depth=4, loss_multiplier_semantic=0).to(dev).train()
grads = parallel.FlatGradAllReduce(model, flatten_params=True)
opt = torch.optim.AdamW(grads.optimizer_params(), lr=0.01, weight_decay=1e-3, fused=True)
batch = bench.make_batch(262144, seed=0, device=dev, trees=1)
def step():
    grads.zero()
    loss, _ = model(batch, return_loss=True)
    (loss * 50).backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if t >= 2 and (e.key.startswith("aten::") or "Mem" in e.key):
        stack = [f for f in e.stack if "pn2_amd" in f or "extracting-tree" in f or "bench" in f or "prof_glue" in f][:3]
        rows.append((t, e.key, e.count, str(e.input_shapes)[:90], " <- ".join(x.split("/")[-1][:60] for x in stack)))
for t, k, c, sh, st in sorted(rows, reverse=True):
    print(f"{t:8.1f}us x{c:<3d} {k:28s} {sh:90s} {st}")
