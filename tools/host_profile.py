"""GPU box: where the HOST time of one headline step goes (cProfile over 20 steps, GPU work left asynchronous).
    python tools/host_profile.py [--dtype bf16]"""
import argparse, cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import mlp, parallel
from pn2_amd.PointNet2.PointNet2 import PointNet2
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--single-thread", action="store_true", help="run the autograd engine on the calling thread, so that the profile shows the backward's Python too")
args = ap.parse_args()
mlp.GEMM_PRECISION = args.dtype
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
    grads.allreduce()
    opt.step()
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {1e3 * (t1 - t0) / 20:.3f} ms; with the final sync {1e3 * (t2 - t0) / 20:.3f} ms")
if args.single_thread:
    torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(args.top)
