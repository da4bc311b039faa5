import torch
torch.manual_seed(0)
N, C = 3000, 256
x0 = torch.randn(N, C, device="cuda")
counts = torch.randint(1, 9, (600,), device="cuda")
counts = counts[: int((counts.cumsum(0) <= N).sum())]
n = int(counts.sum())
x0 = x0[:n]
g = torch.randn(len(counts), C, device="cuda")
junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(16)]; del junk
res = []
for it in range(6):
    x = x0.clone().requires_grad_(True)
    y = torch.segment_reduce(x, "max", lengths=counts, axis=0)
    y.backward(g)
    res.append((y.detach().clone(), x.grad.clone()))
for it in range(1, 6):
    print("fwd diff", float((res[it][0] - res[0][0]).abs().max()), "bwd diff", float((res[it][1] - res[0][1]).abs().max()),
          "nan", int(torch.isnan(res[it][1]).sum()))
# manual
seg = torch.repeat_interleave(torch.arange(len(counts), device="cuda"), counts)
y = res[0][0]
mask = (x0 == y[seg]).float()
ties = torch.segment_reduce(mask, "sum", lengths=counts, axis=0)
man = mask * (g / ties)[seg]
print("manual vs torch bwd", [float((man - r[1]).abs().max()) for r in res], "max ties", float(ties.max()))
