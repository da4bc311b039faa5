"""One training step as ONE HIP graph (extension; the reference has nothing like it).

A step of the hot path is ~250 kernel launches, ~150 of them a few microseconds long; replaying them from a graph
removes the host launch path and most of the gaps between dependent kernels.  Everything in the step is
capturable: the library only enqueues kernels on the stream it is given, workspaces come from torch's graph-private
pool, the loss is sync-free, and the FPS start indices are fed through `StartIndexFeed` (fixed pinned slots,
redrawn before every replay in the reference's RNG order).  Shapes are frozen at capture: use it for fixed-size
batches (monolithic trees, BASELINE configs[1]/[2]); ragged raster mini-batches stay eager.
"""
import torch

from .PointNet2.pointnet2_utils import StartIndexFeed


class GraphedTrainStep:
    """step_fn() -> loss must do the whole step (zero grads, forward, backward, [all-reduce], optimizer) on static
    input tensors; the optimizer must be capturable.  Call the object to run one step; it returns the (static) loss
    tensor of that step."""

    def __init__(self, step_fn, warmup=3):
        self.feed = StartIndexFeed()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), self.feed:
            for _ in range(warmup):
                self.feed.begin_pass()
                step_fn()
                side.synchronize()      # the pinned start-index slots are rewritten by the next pass
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with self.feed:
            with torch.cuda.graph(self.graph):
                self.loss = step_fn()
        self.done = torch.cuda.Event()

    def __call__(self):
        # The replay's host-to-device copy nodes read the pinned slots when they EXECUTE, not when they are enqueued:
        # the previous replay must have consumed its draws before they are overwritten (otherwise two steps share
        # start indices and the reference's RNG order is lost).
        self.done.synchronize()
        self.feed.redraw()
        self.graph.replay()
        self.done.record()
        return self.loss
