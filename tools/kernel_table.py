"""GPU: per-(kernel, launch shape) table of one training step of the bench workload."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from bench import make_batch
from pn2_amd import _hip, parallel
from pn2_amd.PointNet2.PointNet2 import PointNet2
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PointNet2(depth=depth, loss_multiplier_semantic=0).to(dev).train()
grads = parallel.FlatGradAllReduce(model)
opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3)
batch = make_batch(n, 0, dev)
def step():
    grads.zero(); loss, _ = model(batch, return_loss=True); (loss * 50).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
K = 5
def run():
    for _ in range(K): step()
    torch.cuda.synchronize()
groups = _hip.kernel_profile(run)
tot = sum(g["ms"] for g in groups) / K
print(f"sum of kernel time per step: {tot:.3f} ms")
for g in sorted(groups, key=lambda g: -g["ms"])[:int(os.environ.get("PN2_KT_ROWS", "40"))]:
    us = 1e3 * g["ms"] / g["calls"]
    print(f"{g['name']:24s} x{g['calls'] / K:5.1f}  {us:9.1f} us/launch  {g['ms'] / K:7.3f} ms/step  "
          f"{g['bytes'] / us / 1e3:8.1f} GB/s  {g['flops'] / us / 1e6:7.2f} TFLOP/s   bytes={g['bytes']:.3g} flops={g['flops']:.3g}")
