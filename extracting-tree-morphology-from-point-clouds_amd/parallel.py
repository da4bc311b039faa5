"""Data-parallel wiring for the hot path: one process per GPU, one gradient all-reduce per optimizer step.

The reference is single-process (no DDP, no collective anywhere); the path shards on the
batch dimension (trees, or raster mini-batches of one tree) with per-replica BatchNorm statistics, so the only
exchange is the sum of the flat fp32 gradient (983 845 parameters = 3.94 MB at depth 4/5).  xGMI is
point-to-point and the buffer is tiny, so it is sent as ONE RCCL all-reduce per step -- in streaming mode the
model runs ~40 backward passes per step, which is why this is not torch's per-bucket DDP hook: gradients simply
accumulate locally until ``allreduce_gradients`` is called between the last backward and the optimizer step
(the place of reference train_utils.py:57-61).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, local_rank, world_size); a plain single process needs none of the variables."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm.  PN2_DIST_BACKEND=gloo is the rehearsal path (several ranks on one GPU box)
            backend = os.environ.get("PN2_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatGradAllReduce:
    """Owns one contiguous fp32 buffer that every parameter's ``.grad`` is a view of, so the per-step
    exchange is a single collective with no packing copies.

    flatten_params=True additionally re-homes the parameters themselves into one flat buffer (`flat_param`, whose
    ``.grad`` is the flat gradient): an optimizer built over ``[sync.flat_param]`` then updates the whole model with ONE
    fused launch instead of a multi-tensor sweep over ~140 small tensors (95 -> ~10 us per step for AdamW; same update,
    element for element).  Every tensor starts on a 16-byte boundary in both buffers -- unaligned weights would push the
    GEMM operand loads onto the scalar path (that is what sank the round-1 attempt).  state_dict(), load_state_dict() and
    module.to() keep working: the parameters are views."""

    ALIGN = 4   # floats

    def __init__(self, module, broadcast=True, flatten_params=False):
        if broadcast:
            broadcast_module_state(module)
        self.params = [p for p in module.parameters() if p.requires_grad]
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += -(-p.numel() // self.ALIGN) * self.ALIGN
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_param = None
        if flatten_params:
            flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
            with torch.no_grad():
                for p, off in zip(self.params, offs):
                    flat_param[off:off + p.numel()].copy_(p.detach().reshape(-1))
                    p.data = flat_param[off:off + p.numel()].view_as(p)
            self.flat_param = torch.nn.Parameter(flat_param)
            self.flat_param.grad = self.flat
        for p, off in zip(self.params, offs):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def optimizer_params(self, chunks=36):
        """The flat parameter buffer as `chunks` equal leaf tensors (views of it, each with its slice of the flat gradient
        as ``.grad``) for a fused optimizer: torch's multi-tensor kernels hand every tensor to workgroups in 65 536-element
        pieces, so ONE 984 k-element tensor is updated by 16 workgroups (45 us for AdamW on a 256-CU device); 36 slices are
        are still ONE launch (36 is what a four-list multi-tensor launch takes) on 36 workgroups.  Same update, element for element."""
        if self.flat_param is None:
            raise RuntimeError("optimizer_params needs flatten_params=True")
        n = self.flat_param.numel()
        step = -(-n // max(1, chunks))
        step = -(-step // 4) * 4                           # 16-byte aligned slices
        out = []
        for a in range(0, n, step):
            p = torch.nn.Parameter(self.flat_param.data[a:a + step])
            p.grad = self.flat[a:a + step]
            out.append(p)
        return out

    def packed(self):
        """The gradients without the alignment padding, in parameter order (what torch.cat of the .grad tensors gives)."""
        return torch.cat([p.grad.reshape(-1) for p in self.params])

    def zero(self):
        """Replaces optimizer.zero_grad(): keeps the views alive (set_to_none would drop them)."""
        self.flat.zero_()
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() < self.flat.data_ptr():
                raise RuntimeError("a parameter lost its flat gradient view (zero_grad(set_to_none=True)?)")

    def allreduce(self, average=True):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        return self.flat


def broadcast_module_state(module, src=0):
    """Make every replica start from rank `src`'s parameters AND buffers (BatchNorm running statistics, counters):
    replica equality must not depend on every rank having seeded its initialisation identically.  A no-op without
    an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src)


def allreduce_mean(values, device):
    """Mean over ranks of a few host scalars (validation losses): every rank then takes the same early-stopping and
    checkpoint decisions."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return (t / dist.get_world_size()).tolist()


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for this rank (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shards(costs, world):
    """Greedy longest-processing-time assignment of items (e.g. rasters weighted by padded point count) to
    ranks; returns a list of index lists.  Raster sizes of one tree range over 1..6112 points, so contiguous
    sharding would leave ranks idle."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads, out = [0.0] * world, [[] for _ in range(world)]
    for i in order:
        r = loads.index(min(loads))
        out[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in out]
