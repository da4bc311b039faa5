"""Pointwise (1x1-conv) MLP chains on channels-last rows, served by libpn2hip's MFMA chain kernels.

Every MLP of the hot path -- the per-group Conv2d/BatchNorm2d/ReLU stacks + max over K of set abstraction
(reference blocks.py:93-98), the Conv1d/BatchNorm1d/ReLU stacks of feature propagation (:213-215) and the
ConvHead (:7-35) -- is a chain of ``rows x C_in -> rows x C_out`` contractions with per-channel batch statistics
over all rows.  The modules keep the reference's parameter containers (state-dict keys and shapes unchanged)
and hand their layers to ``chain_rows``; one C-ABI call runs the whole chain forward, one runs it backward
(include/pn2_hip.h: pn2_mlp_chain_{fwd,bwd}_f32).
"""
import ctypes
import functools
import os
import threading

import torch

from . import _hip


def _bn_momentum(bn, bump):
    """nn.BatchNorm bookkeeping on the host side: num_batches_tracked += 1 (queued in `bump`: one fused launch per chain
    instead of one per layer), resolve momentum=None."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn.momentum is None:                       # cumulative average: the new count is needed right now
            bn.num_batches_tracked.add_(1)
            return 1.0 / float(bn.num_batches_tracked)
        bump.append(bn.num_batches_tracked)
    return 0.0 if bn.momentum is None else float(bn.momentum)


# The chain's backward kernels ACCUMULATE into their dweight / dbias / dgamma / dbeta targets.  With this switch on they
# are pointed straight at the parameters' `.grad` buffers (created zeroed on first use) and the Function returns None
# for those inputs: no zero-filled temporaries and no AccumulateGrad `add_` per parameter and backward -- ~190 tiny
# launches per step of the depth-4 model, per mini-batch in streaming mode.  Same values (x + 0 first, then the same
# running sum).  Turn it off for torch.autograd.grad()-style use, which expects the gradients to be RETURNED.
FUSED_GRAD_ACCUMULATION = True

# Precision of the large contractions (include/pn2_hip.h: PN2_PRECISION_*).  "f32" is the parity mode (exact fp32 MFMA, the
# reference's width); "bf16" rounds the MFMA operands to bfloat16 with fp32 accumulation -- a separate throughput mode with
# its own tolerance (tests/test_bf16_mode.py, bench.py --dtype bf16).  Read when a chain is called; the backward of a chain
# uses the precision its forward ran with.
GEMM_PRECISION = "f32"
_PRECISION_CODE = {"f32": 0, "bf16": 1}


def _coop(dev):
    """The deep levels' chains run as one persistent launch per direction when the caller hands over arrival counters
    (include/pn2_hip.h: pn2_coop); the library decides per call (PN2_NO_COOP=1 switches it off there)."""
    from . import ops
    return ops.coop_ctl(dev)


def _grad_target(leaf, value, need, dev):
    """-> (tensor the kernels accumulate into or None, gradient to return to autograd or None)."""
    if value is None or not need:
        return None, None
    if FUSED_GRAD_ACCUMULATION and leaf is not None and leaf.is_leaf and leaf.requires_grad:
        if leaf.grad is None:
            leaf.grad = torch.zeros_like(leaf, memory_format=torch.contiguous_format)
        gr = leaf.grad
        if gr.is_contiguous() and gr.dtype == torch.float32 and gr.device == dev and gr.numel() == value.numel():
            return gr, None
    z = torch.zeros_like(value)
    return z, z


class _ChainFn(torch.autograd.Function):
    """forward(x [R,Cin], meta, *params) -> out; params = (weight, bias, gamma, beta) per layer (None allowed)."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        _hip.require_device(x)
        lib = _hip.lib()
        link = meta.get("link")                         # LazyRows of the producing chain: x is its pre-activation rows
        if not (link is not None and x.dtype == torch.bfloat16):   # (a linked producer's rows may be bfloat16: STORE_BF16 below)
            x = _hip.f32(x)
        if x.stride(1) != 1:
            x = x.contiguous()
        rows, cin0 = x.shape
        n = len(meta["layers"])
        pool_k, training = meta["pool_k"], meta["training"]
        # row segments (whole-tree execution): per-mini-batch BatchNorm statistics inside one chain call
        seg_ptr, seg_keep = _hip.segments_arg(meta.get("seg_off") if training else None)
        nseg = 1 if seg_ptr is None else len(meta["seg_off"]) - 1
        dev = x.device
        arr = (_hip.MLPLayer * n)()
        if link is not None:
            arr[0].in_stats, arr[0].in_relu = link.stats.data_ptr(), int(link.relu)
        lazy_out = bool(meta.get("lazy_out"))
        precision = _PRECISION_CODE[GEMM_PRECISION]
        # bf16 mode: the large chains keep their pre-BatchNorm rows (and gradient rows) as bfloat16 in memory -- the library says
        # which chains (include/pn2_hip.h: PN2_CHAIN_STORE_BF16); PN2_BF16_STORAGE=0 keeps fp32 rows (round 2's bf16 mode)
        cin = cin0
        for i, spec in enumerate(meta["layers"]):
            arr[i].cin, arr[i].cout, arr[i].has_bn = cin, spec["cout"], int(spec["has_bn"])
            cin = spec["cout"]
        store16 = (GEMM_PRECISION == "bf16" and training and os.environ.get("PN2_BF16_STORAGE", "1") != "0"
                   and (link is None or x.dtype == torch.bfloat16)          # (linked to fp32 rows: stay fp32)
                   and bool(lib.pn2_mlp_chain_bf16_storage(rows, arr, n, int(pool_k))))
        if x.dtype == torch.bfloat16 and not store16:   # a bfloat16 producer feeding a chain that keeps fp32 rows
            x = x.float()
        ydtype = torch.bfloat16 if store16 else torch.float32
        ys, stats = [], []
        cin = cin0
        for i, spec in enumerate(meta["layers"]):
            w, b, g, be = params[4 * i:4 * i + 4]
            cout = spec["cout"]
            L = arr[i]
            L.cin, L.cout = cin, cout
            L.weight, L.bias = w.data_ptr(), _hip.ptr(b)
            L.has_bn, L.relu = int(spec["has_bn"]), int(spec["relu"])
            L.gamma, L.beta = _hip.ptr(g), _hip.ptr(be)
            L.running_mean, L.running_var = _hip.ptr(spec["running_mean"]), _hip.ptr(spec["running_var"])
            L.eps, L.momentum = spec["eps"], spec["momentum"]
            last = i == n - 1
            if last and not spec["has_bn"] and pool_k <= 1:
                y = None                                          # the chain writes `out` directly
            else:
                y = torch.empty(rows, cout, dtype=ydtype if spec["has_bn"] else torch.float32, device=dev)
            st = torch.empty(8 * nseg, cout, dtype=torch.float32, device=dev) if spec["has_bn"] else None
            L.y, L.stats = _hip.ptr(y), _hip.ptr(st)
            ys.append(y)
            stats.append(st)
            cin = cout
        cout_last = cin
        if pool_k > 1:
            out = torch.empty(rows // pool_k, cout_last, dtype=torch.float32, device=dev)
            arg = torch.empty(rows // pool_k, cout_last, dtype=torch.int32, device=dev)
        elif lazy_out:
            out, arg = None, None                       # the consumer reads ys[-1] through stats[-1]
        else:
            out = torch.empty(rows, cout_last, dtype=torch.float32, device=dev)
            arg = None
        ws = torch.empty(lib.pn2_mlp_workspace_bytes(rows, arr, n, nseg), dtype=torch.uint8, device=dev)
        if store16:
            precision |= _hip.CHAIN_STORE_BF16 | (_hip.CHAIN_X_BF16 if x.dtype == torch.bfloat16 else 0)
        flops = 2 * rows * sum(int(a.cin) * int(a.cout) for a in arr)
        nbytes = 4 * rows * (cin0 + 2 * sum(int(a.cout) for a in arr))
        _hip.call("mlp_chain_fwd", lib.pn2_mlp_chain_fwd_f32, x.data_ptr(), x.stride(0), rows, arr, n, int(training),
                  int(pool_k), _hip.ptr(out), _hip.ptr(arg), seg_ptr, precision | (_hip.CHAIN_LAZY_OUT if lazy_out else 0),
                  _coop(dev), ws.data_ptr(), ws.numel(), _hip.stream_ptr(), nbytes=nbytes, flops=flops)
        ctx.meta = meta
        ctx.nseg = nseg
        ctx.precision = precision
        ctx.arr = arr
        ctx.dims = (rows, cin0)
        ctx.save_for_backward(x, arg, *[t for t in ys if t is not None], *[t for t in stats if t is not None], *params)
        ctx.layout = ([t is not None for t in ys], [t is not None for t in stats])
        ctx.link = link                                  # keeps the producer's coefficient blocks alive for the backward
        if lazy_out:
            # what leaves is the last layer's PRE-activation rows; the gradient that comes back for it is the gradient with
            # respect to the activated rows (chain_pair_rows / chain_rows(link=...) are the only consumers): see LazyRows
            ctx.mark_non_differentiable(stats[-1])
            return ys[-1], stats[-1]
        return out

    @staticmethod
    def backward(ctx, dout, *_):
        lib = _hip.lib()
        meta = ctx.meta
        if not meta["training"] and any(sp["has_bn"] for sp in meta["layers"]):
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the hot path")
        rows, cin0 = ctx.dims
        n = len(meta["layers"])
        saved = list(ctx.saved_tensors)
        x, arg = saved[0], saved[1]
        has_y, has_st = ctx.layout
        pos = 2
        ys = []
        for h in has_y:
            ys.append(saved[pos] if h else None)
            pos += int(h)
        stats = []
        for h in has_st:
            stats.append(saved[pos] if h else None)
            pos += int(h)
        params = saved[pos:]
        dev = x.device
        dout = dout.contiguous()
        store16 = bool(ctx.precision & _hip.CHAIN_STORE_BF16)
        if dout.dtype == torch.bfloat16 and not store16:
            dout = dout.float()
        last = meta["layers"][-1]
        if store16 and dout.dtype != torch.bfloat16 and last["has_bn"]:
            dout = dout.to(torch.bfloat16)                 # a bfloat16 chain takes its upstream gradient as bfloat16 rows (the
                                                           # narrow last layer of a head reads its few fp32 columns itself)
        arr = ctx.arr                                      # the forward's layer table (shapes, weights, y, stats)
        grads = []
        cin, maxc = cin0, 0
        for i, spec in enumerate(meta["layers"]):
            w, b, g, be = params[4 * i:4 * i + 4]
            L = arr[i]
            need = ctx.needs_input_grad[2 + 4 * i:2 + 4 * i + 4]
            leaves = spec.get("leaves", (None, None, None, None))
            # (a bias of a conv that feeds a train-mode BatchNorm has an identically zero gradient: its target only has
            # to exist)
            tw, dw = _grad_target(leaves[0], w, need[0], dev)
            tb, db = _grad_target(leaves[1], b, need[1], dev)
            tg, dg = _grad_target(leaves[2], g, need[2], dev)
            tbe, dbe = _grad_target(leaves[3], be, need[3], dev)
            L.dweight = _hip.ptr(tw)
            L.dbias = None if spec["has_bn"] else _hip.ptr(tb)
            L.dgamma, L.dbeta = _hip.ptr(tg), _hip.ptr(tbe)
            grads += [dw, db, dg, dbe]
            if i > 0:
                maxc = max(maxc, cin)
            cin = spec["cout"]
        maxc = max(maxc, cin)
        skip = int(meta.get("dx_first_col", 0))
        # columns in front of `skip` are never computed: zero them so that the tensor handed to autograd is defined
        into = getattr(ctx, "dx_into", None)              # chain_pair_rows: add to the sibling chain's input gradient
        flags = ctx.precision
        if into is not None:
            dx, flags = into, flags | _hip.CHAIN_ACCUMULATE_DX
        else:
            dx = torch.empty(rows, cin0, dtype=x.dtype, device=dev) if ctx.needs_input_grad[0] else None   # (bfloat16 for a bfloat16 producer)
            if dx is not None and skip:
                flags |= _hip.CHAIN_ZERO_LEAD
        if store16:
            flags |= (_hip.CHAIN_DOUT_BF16 if dout.dtype == torch.bfloat16 else 0) | \
                     (_hip.CHAIN_DX_BF16 if dx is not None and dx.dtype == torch.bfloat16 else 0)
        sdt = torch.bfloat16 if store16 else torch.float32
        sa = torch.empty(rows * maxc, dtype=sdt, device=dev)
        sb = torch.empty(rows * maxc, dtype=sdt, device=dev)
        seg_ptr, seg_keep = _hip.segments_arg(meta.get("seg_off") if ctx.nseg > 1 else None)
        ws = torch.empty(lib.pn2_mlp_workspace_bytes(rows, arr, n, ctx.nseg), dtype=torch.uint8, device=dev)
        flops = 4 * rows * sum(int(a.cin) * int(a.cout) for a in arr)
        nbytes = 4 * rows * (cin0 + 5 * sum(int(a.cout) for a in arr))
        # Deferral is safe only when every weight-gradient target of this call is the parameter's OWN .grad buffer (nothing is
        # returned to autograd for it): a returned tensor would be read by AccumulateGrad / summed with other uses of the
        # weight BEFORE the end-of-pass reduction has written into it.
        targets_are_grad_buffers = all(g is None for g in grads[0::4])
        deferred = _DeferredWgrad.enlist(ws) if targets_are_grad_buffers else None
        # linked chains: BatchNorm-backward column sums handed from the consumer's dgrad epilogue to the producer
        handed = (meta.get("lazy_handle") or {}).pop("partial", None)
        if handed is not None:                            # this chain produced a LazyRows and its consumer did the sums
            arr[n - 1].out_partial, arr[n - 1].out_partial_rows, arr[n - 1].out_partial_cpb = handed[0].data_ptr(), handed[1], handed[2]
        else:
            arr[n - 1].out_partial = None
        emit = None
        link = getattr(ctx, "link", None)
        if (link is not None and dx is not None and skip == 0 and getattr(ctx, "dx_is_complete", True)
                and not link.handle.get("shared") and not os.environ.get("PN2_NO_LINK_SUMS")):
            r_, c_ = ctypes.c_int32(0), ctypes.c_int32(0)
            nb = lib.pn2_mlp_link_partial_bytes(rows, cin0, ctx.nseg, ctypes.byref(r_), ctypes.byref(c_))
            emit = (torch.empty(nb, dtype=torch.uint8, device=dev), int(r_.value), int(c_.value))
            arr[0].in_partial = emit[0].data_ptr()
        else:
            arr[0].in_partial = None
        _hip.call("mlp_chain_bwd", lib.pn2_mlp_chain_bwd_f32, x.data_ptr(), x.stride(0), rows, arr, n, int(meta["pool_k"]),
                  dout.data_ptr(), _hip.ptr(arg), _hip.ptr(dx), cin0, skip, sa.data_ptr(), sb.data_ptr(), seg_ptr, flags,
                  deferred, _coop(dev), ws.data_ptr(), ws.numel(), _hip.stream_ptr(), nbytes=nbytes, flops=flops)
        if emit is not None:
            link.handle["partial"] = emit
        if getattr(ctx, "keep_layer0", False):        # _ChainPairFn: the fused first-layer dgrad of the two heads needs them
            ctx.kept = {"dz0": sa, "scratch": sb, "arr": arr, "x": x, "flags": ctx.precision, "seg": (seg_ptr, seg_keep), "ws": ws}
        return (dx, None, *grads)


class _DeferredWgrad:
    """Weight-gradient slab reductions of a whole backward pass in as few launches as possible: every chain backward appends
    the reductions it owes to a host list (pn2_wgrad_tasks, include/pn2_hip.h) and leaves its slabs in its workspace; the
    reductions run from an autograd-engine callback when the pass has finished -- before loss.backward() returns, so nobody
    sees a weight gradient without them.  One record PER BACKWARD PASS (keyed by the engine's graph-task id) and device: a
    nested or re-entrant backward (checkpointing, a backward inside a hook) has its own list and its own callback and never
    touches the outer pass's.  The library itself holds no state.  A pass that died leaves its record behind; records are few
    (the workspaces they pin are dropped when more than MAX_LIVE passes are pending).  PN2_NO_DEFER_WGRAD=1 reduces per call."""
    MAX_LIVE = 8
    _lock = threading.Lock()
    _passes = {}          # (graph task id, device index) -> {"lists": [WgradTasks], "keep": [workspaces], "stream": ptr}

    @classmethod
    def enabled(cls):
        return not os.environ.get("PN2_NO_DEFER_WGRAD") and hasattr(torch._C, "_current_graph_task_id")

    @classmethod
    def enlist(cls, ws):
        """-> ctypes pointer to the list the calling chain backward appends to, or None (reduce inside the call)."""
        if not cls.enabled():
            return None
        tid = torch._C._current_graph_task_id()
        if tid < 0:
            return None
        key = (tid, ws.device.index)
        with cls._lock:
            rec = cls._passes.get(key)
            if rec is None:
                while len(cls._passes) >= cls.MAX_LIVE:          # leftovers of passes that died
                    cls._passes.pop(next(iter(cls._passes)))
                rec = cls._passes[key] = {"lists": [_hip.WgradTasks()], "keep": [], "stream": _hip.stream_ptr()}
                torch.autograd.Variable._execution_engine.queue_callback(functools.partial(cls.flush, key))
            if rec["lists"][-1].n > _hip.WGRAD_TASKS_MAX - 8:    # room for the longest chain
                rec["lists"].append(_hip.WgradTasks())
            rec["keep"].append(ws)
            return ctypes.pointer(rec["lists"][-1])

    @classmethod
    def flush(cls, key):
        with cls._lock:
            rec = cls._passes.pop(key, None)
        if rec is None:
            return
        n = sum(int(L.n) for L in rec["lists"])
        if n == 0:
            return
        arr = (_hip.WgradTask * n)()
        k = 0
        for L in rec["lists"]:
            for i in range(int(L.n)):
                arr[k] = L.t[i]
                k += 1
        st = _hip.lib().pn2_mlp_reduce_wgrad(arr, n, rec["stream"])
        if st != 0:
            raise RuntimeError(f"pn2_mlp_reduce_wgrad -> {st}")


class _Sub:
    """Stands in for an autograd context when _ChainFn's forward / backward run as plain functions inside _ChainPairFn."""

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _ChainPairFn(torch.autograd.Function):
    """Two chains on the SAME input rows as one autograd node: forward(x, meta_a, meta_b, n_a, *params_a, *params_b) ->
    (out_a, out_b).  The backward runs both chains' backward passes and lets the second one ADD its input gradient to the
    first one's (PN2_CHAIN_ACCUMULATE_DX) -- autograd would otherwise sum the two [rows, cin] tensors with a separate
    kernel (65 us at 262144 x 128).  Same values: a + b either way."""

    @staticmethod
    def forward(ctx, x, meta_a, meta_b, n_a, *params):
        a, b = _Sub(), _Sub()
        out_a = _ChainFn.forward(a, x, meta_a, *params[:n_a])
        out_b = _ChainFn.forward(b, x, meta_b, *params[n_a:])
        ctx.subs = (a, b, n_a)
        return out_a, out_b

    @staticmethod
    def backward(ctx, da, db):
        a, b, n_a = ctx.subs
        need = ctx.needs_input_grad
        if need[0] and _ChainPairFn._fusable(a, b):
            return _ChainPairFn._backward_fused(ctx, a, b, n_a, da, db)
        a.needs_input_grad = (need[0], False) + tuple(need[4:4 + n_a])
        b.needs_input_grad = (need[0], False) + tuple(need[4 + n_a:])
        a.dx_is_complete = False                       # the second chain adds to it: only that one can do the producer's sums
        ga = _ChainFn.backward(a, da)
        b.dx_into = ga[0]
        gb = _ChainFn.backward(b, db)
        ctx.subs = None
        return (ga[0], None, None, None) + tuple(ga[2:]) + tuple(gb[2:])


def _pair_fusable(a, b):
    """Both chains: two layers, a BatchNorm after the first, equal shapes, big enough for the 128-row tiles (the conditions
    of pn2_mlp_pair_dgrad_f32), the same row storage."""
    if os.environ.get("PN2_NO_PAIR_DGRAD"):
        return False
    ma, mb = a.meta, b.meta
    if len(ma["layers"]) != 2 or len(mb["layers"]) != 2 or ma["pool_k"] != 1 or mb["pool_k"] != 1:
        return False
    la, lb = a.arr[0], b.arr[0]
    rows, cin = a.dims
    return (bool(la.has_bn) and bool(lb.has_bn) and int(la.cin) == int(lb.cin) == 128 and int(la.cout) == int(lb.cout)
            and int(la.cout) % 16 == 0 and int(la.relu) == int(lb.relu) and rows >= 12288 and a.dims == b.dims
            and a.precision == b.precision and ma.get("seg_off") == mb.get("seg_off") and ma["training"] and mb["training"])


def _pair_backward_fused(ctx, a, b, n_a, da, db):
    """Both chains' backward passes without their first-layer dgrad, then dx = dY_a W_a + dY_b W_b as ONE contraction
    (include/pn2_hip.h: pn2_mlp_pair_dgrad_f32), which also leaves the producer's BatchNorm-backward sums."""
    lib = _hip.lib()
    need = ctx.needs_input_grad
    a.needs_input_grad = (False, False) + tuple(need[4:4 + n_a])
    b.needs_input_grad = (False, False) + tuple(need[4 + n_a:])
    a.keep_layer0 = b.keep_layer0 = True
    ga = _ChainFn.backward(a, da)
    gb = _ChainFn.backward(b, db)
    ka, kb = a.kept, b.kept
    x = ka["x"]
    rows, cin0 = a.dims
    dev = x.device
    dx = torch.empty(rows, cin0, dtype=x.dtype, device=dev)
    flags = a.precision
    store16 = bool(flags & _hip.CHAIN_STORE_BF16)
    if store16:
        flags |= (_hip.CHAIN_X_BF16 if x.dtype == torch.bfloat16 else 0) | (_hip.CHAIN_DX_BF16 if dx.dtype == torch.bfloat16 else 0)
    la, lb = ka["arr"][0], kb["arr"][0]
    link = getattr(a, "link", None)
    emit = None
    if link is not None and not link.handle.get("shared") and not os.environ.get("PN2_NO_LINK_SUMS"):
        r_, c_ = ctypes.c_int32(0), ctypes.c_int32(0)
        nb = lib.pn2_mlp_link_partial_bytes(rows, cin0, a.nseg, ctypes.byref(r_), ctypes.byref(c_))
        emit = (torch.empty(nb, dtype=torch.uint8, device=dev), int(r_.value), int(c_.value))
        la.in_partial = emit[0].data_ptr()
    else:
        la.in_partial = None
    cout = int(la.cout)
    _hip.call("mlp_pair_dgrad", lib.pn2_mlp_pair_dgrad_f32, rows, ctypes.byref(la), ka["dz0"].data_ptr(), ctypes.byref(lb),
              kb["dz0"].data_ptr(), x.data_ptr(), x.stride(0), dx.data_ptr(), cin0, ka["seg"][0], flags, _hip.stream_ptr(),
              nbytes=4 * rows * (cin0 + 4 * cout), flops=4 * rows * cin0 * cout)
    if emit is not None:
        link.handle["partial"] = emit
    ctx.subs = None
    a.kept = b.kept = None
    return (dx, None, None, None) + tuple(ga[2:]) + tuple(gb[2:])


_ChainPairFn._fusable = staticmethod(_pair_fusable)
_ChainPairFn._backward_fused = staticmethod(_pair_backward_fused)


class batched_counters:
    """Context: collect the `num_batches_tracked` bumps of every chain called inside and apply them with ONE
    torch._foreach_add_ at exit (a forward pass of the depth-4 model otherwise issues ten of those tiny launches)."""
    active = None

    def __enter__(self):
        self.outer, self.pending = batched_counters.active, []
        batched_counters.active = self
        return self

    def __exit__(self, *exc):
        batched_counters.active = self.outer
        if self.pending:
            by_inc = {}
            for t, inc in self.pending:
                by_inc.setdefault(inc, []).append(t)
            for inc, ts in by_inc.items():
                torch._foreach_add_(ts, inc)


class LazyRows:
    """Output of chain_rows(..., lazy_out=True): the last layer's pre-activation rows `y` [R, C] and its BatchNorm
    coefficient block(s) `stats`; the activated rows relu(bn(y)) are never written.  Only a LINKED chain consumes it
    (chain_rows / chain_pair_rows given a LazyRows as x): it applies the activation while staging its operand, and hands
    back the gradient with respect to the activated rows -- which is what the producing chain's backward takes as dout."""

    def __init__(self, y, stats, relu, seg_off, handle=None):
        self.y, self.stats, self.relu, self.seg_off = y, stats, relu, seg_off
        # shared with the producing chain's backward: the consumer leaves the BatchNorm-backward sums of the producing layer
        # here (handle["partial"] = (buffer, block rows, chunks per block)) when it has computed the complete gradient
        self.handle = handle if handle is not None else {}
        self.consumed = False

    @property
    def shape(self):
        return self.y.shape


class _InterpBNFn(torch.autograd.Function):
    """forward(q [B,S,C], idx32 [rows,3], w [rows,3], gamma, beta, bias, meta) -> (y [rows,C] pre-BatchNorm rows, stats): the
    rows of a feature-propagation level whose first convolution was applied to the SAMPLED rows (conv(interp(P)) =
    interp(conv(P)), include/pn2_hip.h "first convolution HOISTED"), interpolated and given their train-mode BatchNorm
    statistics by one launch.  The pair is a LazyRows: the next chain links to it; what comes back is the gradient with
    respect to relu(bn(y)), and the backward scatters dZ(dout, y) to dq without storing it.  meta["rc"]: the dense side is
    ragged clouds (ops.RaggedClouds), otherwise B clouds of rows / B points."""

    @staticmethod
    def forward(ctx, q, idx32, w, gamma, beta, bias, meta):
        # (bias: the hoisted conv's bias, already inside q -- here only so that it gets the exactly zero gradient a bias in
        # front of a train-mode BatchNorm has, instead of the rounding noise of summing dq)
        _hip.require_device(q)
        lib = _hip.lib()
        from . import ops
        B, S, C = q.shape
        rc = meta.get("rc")
        rows = rc.rows if rc is not None else idx32.numel() // 3
        N = rc.n_max if rc is not None else rows // B
        dev = q.device
        q = _hip.f32(q).contiguous()
        training = bool(meta["training"])
        seg_ptr, seg_keep = _hip.segments_arg(meta.get("seg_off") if training else None)
        nseg = 1 if seg_ptr is None else len(meta["seg_off"]) - 1
        y16 = bool(meta.get("rows_bf16")) and training  # bf16 mode with bfloat16 storage: the linked chain keeps bfloat16 rows
        y = torch.empty(rows, C, dtype=torch.bfloat16 if y16 else torch.float32, device=dev)
        st = torch.empty(8 * nseg, C, dtype=torch.float32, device=dev)
        L = _hip.MLPLayer()
        L.cin = L.cout = C
        L.has_bn, L.relu = 1, int(meta["relu"])
        L.gamma, L.beta = _hip.ptr(gamma), _hip.ptr(beta)
        L.running_mean, L.running_var = _hip.ptr(meta["running_mean"]), _hip.ptr(meta["running_var"])
        L.eps, L.momentum = meta["eps"], meta["momentum"]
        L.y, L.stats = y.data_ptr(), st.data_ptr()
        ws = torch.empty(lib.pn2_interp_bn_workspace_bytes(B, rows, S, C, nseg), dtype=torch.uint8, device=dev)
        _hip.call("interp_bn_fwd", lib.pn2_interp_bn_fwd_f32, q.data_ptr(), idx32.data_ptr(), w.data_ptr(),
                  None if rc is None else rc.coff.data_ptr(), None if rc is None else rc.row_cloud.data_ptr(), B, N, S, rows,
                  ctypes.byref(L), seg_ptr, int(training), int(y16), ops.status_word(dev).data_ptr(), ws.data_ptr(), ws.numel(),
                  _hip.stream_ptr(), nbytes=rows * (36 + (2 if y16 else 4) * C) + 4 * B * S * C)
        ctx.save_for_backward(idx32, w, y, st, gamma, beta, bias)
        ctx.meta, ctx.layer, ctx.dims = meta, L, (B, N, S, C, nseg, rows)
        ctx.mark_non_differentiable(st)
        return y, st

    @staticmethod
    def backward(ctx, dout, _):
        lib = _hip.lib()
        if not ctx.meta["training"]:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the hot path")
        idx32, w, y, st, gamma, beta, bias = ctx.saved_tensors
        B, N, S, C, nseg, rows = ctx.dims
        meta, L, dev = ctx.meta, ctx.layer, dout.device
        rc = meta.get("rc")
        y16 = y.dtype == torch.bfloat16
        dout = (dout if dout.dtype == torch.bfloat16 else dout.to(torch.bfloat16)) if y16 else _hip.f32(dout)
        dout = dout.contiguous()
        leaves = meta.get("leaves", (None, None, None))
        tg, dg = _grad_target(leaves[0], gamma, ctx.needs_input_grad[3], dev)
        tbe, dbe = _grad_target(leaves[1], beta, ctx.needs_input_grad[4], dev)
        _, dbias = _grad_target(leaves[2], bias, bias is not None and ctx.needs_input_grad[5], dev)   # zeros
        L.dgamma, L.dbeta = _hip.ptr(tg), _hip.ptr(tbe)
        handed = meta["lazy_handle"].pop("partial", None)      # the linked consumer did the BatchNorm-backward sums
        if handed is not None:
            L.out_partial, L.out_partial_rows, L.out_partial_cpb = handed[0].data_ptr(), handed[1], handed[2]
        else:
            L.out_partial = None
        dq = torch.empty(B, S, C, dtype=torch.float32, device=dev)
        seg_ptr, seg_keep = _hip.segments_arg(meta.get("seg_off"))
        ws = torch.empty(lib.pn2_interp_bn_workspace_bytes(B, rows, S, C, nseg), dtype=torch.uint8, device=dev)
        _hip.call("interp_bn_bwd", lib.pn2_interp_bn_bwd_f32, dout.data_ptr(), idx32.data_ptr(), w.data_ptr(),
                  None if rc is None else rc.coff.data_ptr(), B, N, S, rows, ctypes.byref(L), dq.data_ptr(), seg_ptr, int(y16),
                  ws.data_ptr(), ws.numel(), _hip.stream_ptr(), nbytes=rows * (36 + (4 if y16 else 8) * C) + 4 * B * S * C)
        return dq, None, None, dg, dbe, dbias, None


class _GroupBNFn(torch.autograd.Function):
    """forward(gf [B,N,C], xyz [B,N,3], new_xyz [B,S,3], idx32 [B,S,K], weight [C, 3+D] view of the level's first conv, gamma,
    beta, bias, meta) -> (y [B*S*K, C] pre-BatchNorm rows, stats): the grouped rows of a set-abstraction level whose first
    conv was applied to the SOURCE points (include/pn2_hip.h "Set abstraction with the first convolution HOISTED"):
    y = gf[idx] + W_x (xyz[idx] - centre), with the layer's train-mode BatchNorm statistics from the same launch.  The pair is
    a LazyRows for the chain of the remaining layers; the backward scatters dZ(dout, y) to dgf and sums the coordinate
    weights' gradient.  meta["xcol"]: first coordinate column of the weight (0, or D in the multi-scale channel order)."""

    @staticmethod
    def forward(ctx, gf, xyz, new_xyz, idx32, weight, gamma, beta, bias, meta):
        _hip.require_device(gf)
        lib = _hip.lib()
        from . import ops
        B, N, C = gf.shape
        _, S, K = idx32.shape
        rows, dev = B * S * K, gf.device
        gf = _hip.f32(gf).contiguous()
        xyz, new_xyz = _hip.f32(xyz), _hip.f32(new_xyz).contiguous()
        weight = weight if weight.stride(1) == 1 else weight.contiguous()
        training = bool(meta["training"])
        seg_ptr, seg_keep = _hip.segments_arg(meta.get("seg_off") if training else None)
        nseg = 1 if seg_ptr is None else len(meta["seg_off"]) - 1
        y = torch.empty(rows, C, dtype=torch.float32, device=dev)
        st = torch.empty(8 * nseg, C, dtype=torch.float32, device=dev)
        L = _hip.MLPLayer()
        L.cin = L.cout = C
        L.has_bn, L.relu = 1, int(meta["relu"])
        L.gamma, L.beta = _hip.ptr(gamma), _hip.ptr(beta)
        L.running_mean, L.running_var = _hip.ptr(meta["running_mean"]), _hip.ptr(meta["running_var"])
        L.eps, L.momentum = meta["eps"], meta["momentum"]
        L.y, L.stats = y.data_ptr(), st.data_ptr()
        ws = torch.empty(lib.pn2_group_bn_workspace_bytes(B, S, K, C, nseg), dtype=torch.uint8, device=dev)
        xs = (xyz.stride(0), xyz.stride(1), xyz.stride(2))
        _hip.call("group_bn_fwd", lib.pn2_group_bn_fwd_f32, gf.data_ptr(), xyz.data_ptr(), *xs, new_xyz.data_ptr(), idx32.data_ptr(),
                  weight.data_ptr() + 4 * int(meta["xcol"]), weight.stride(0), B, N, S, K, ctypes.byref(L), seg_ptr, int(training),
                  ops.status_word(dev).data_ptr(), ws.data_ptr(), ws.numel(), _hip.stream_ptr(),
                  nbytes=rows * (8 * C + 20), flops=6 * rows * C)
        ctx.save_for_backward(xyz, new_xyz, idx32, y, st, gamma, beta, bias)
        ctx.meta, ctx.layer, ctx.dims, ctx.wshape = meta, L, (B, N, S, K, C, nseg), tuple(weight.shape)
        ctx.mark_non_differentiable(st)
        return y, st

    @staticmethod
    def backward(ctx, dout, _):
        lib = _hip.lib()
        if not ctx.meta["training"]:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the hot path")
        xyz, new_xyz, idx32, y, st, gamma, beta, bias = ctx.saved_tensors
        B, N, S, K, C, nseg = ctx.dims
        meta, L, dev = ctx.meta, ctx.layer, dout.device
        rows = B * S * K
        dout = _hip.f32(dout).contiguous()
        leaves = meta.get("leaves", (None, None, None))
        tg, dg = _grad_target(leaves[0], gamma, ctx.needs_input_grad[5], dev)
        tbe, dbe = _grad_target(leaves[1], beta, ctx.needs_input_grad[6], dev)
        _, dbias = _grad_target(leaves[2], bias, bias is not None and ctx.needs_input_grad[7], dev)   # zeros
        L.dgamma, L.dbeta = _hip.ptr(tg), _hip.ptr(tbe)
        handed = meta["lazy_handle"].pop("partial", None)
        if handed is not None:
            L.out_partial, L.out_partial_rows, L.out_partial_cpb = handed[0].data_ptr(), handed[1], handed[2]
        else:
            L.out_partial = None
        dgf = torch.empty(B, N, C, dtype=torch.float32, device=dev)
        dw = torch.zeros(ctx.wshape, dtype=torch.float32, device=dev)      # only the three coordinate columns are written
        seg_ptr, seg_keep = _hip.segments_arg(meta.get("seg_off"))
        ws = torch.empty(lib.pn2_group_bn_workspace_bytes(B, S, K, C, nseg), dtype=torch.uint8, device=dev)
        xs = (xyz.stride(0), xyz.stride(1), xyz.stride(2))
        _hip.call("group_bn_bwd", lib.pn2_group_bn_bwd_f32, dout.data_ptr(), xyz.data_ptr(), *xs, new_xyz.data_ptr(), idx32.data_ptr(),
                  B, N, S, K, ctypes.byref(L), dgf.data_ptr(), dw.data_ptr() + 4 * int(meta["xcol"]), dw.stride(0), seg_ptr,
                  ws.data_ptr(), ws.numel(), _hip.stream_ptr(), nbytes=rows * (8 * C + 20) + 4 * B * N * C, flops=8 * rows * C)
        return dgf, None, None, None, dw, dg, dbe, dbias, None


HOIST_WIDTHS = (64, 128, 256)
HOIST_GROUP_WIDTHS = (32, 64, 128, 256)
# a set-abstraction level is hoisted when it has at least this many grouped rows (smaller levels run as one cooperative
# launch, which a linked chain cannot) and more grouped rows than source points
HOIST_GROUP_MIN_ROWS = 32768


def interp_bn_rows(q, idx32, w, bn, relu=True, seg_off=None, bias=None, rc=None):
    """q [B,S,C] = the level's first conv applied to the sampled rows; idx32 / w: three_nn of the dense points ->
    LazyRows of relu(bn(interp(q))) over the B*N dense rows (train-mode `bn`, C in HOIST_WIDTHS); see _InterpBNFn.
    bias: that conv's bias parameter when q was computed with a DETACHED copy of it (hoisted_conv): it gets a zero gradient.
    rc: the dense side as ops.RaggedClouds (idx32 / w packed rows) instead of B equal clouds; seg_off: row segments made of
    whole clouds, as for chain_rows."""
    meta = _bn_meta(bn, relu, seg_off, (bn.weight, bn.bias, bias))
    meta["rc"] = rc
    # bf16 mode: bfloat16 rows where the chains keep theirs (the full-resolution 128-wide chains, include/pn2_hip.h
    # PN2_CHAIN_STORE_BF16) -- the linked consumer then runs on bfloat16 rows end to end
    meta["rows_bf16"] = (GEMM_PRECISION == "bf16" and os.environ.get("PN2_BF16_STORAGE", "1") != "0" and q.shape[2] == 128
                         and idx32.numel() // 3 >= 32768)
    y, st = _InterpBNFn.apply(q, idx32, w, bn.weight, bn.bias, bias, meta)
    return LazyRows(y, st, bool(relu), meta.get("seg_off"), meta["lazy_handle"])


def _bn_meta(bn, relu, seg_off, leaves):
    """BatchNorm bookkeeping shared by the hoisted-layer functions -> meta; bumps num_batches_tracked."""
    bump = []
    meta = {"relu": bool(relu), "eps": float(bn.eps), "momentum": _bn_momentum(bn, bump), "running_mean": None,
            "running_var": None, "leaves": leaves, "lazy_handle": {}, "training": bn.training or bn.running_mean is None}
    if bn.track_running_stats and bn.running_mean is not None:
        meta["running_mean"], meta["running_var"] = bn.running_mean, bn.running_var
    nseg = 1
    if seg_off is not None and len(seg_off) > 2:
        if bn.momentum is None:
            raise NotImplementedError("row segments with cumulative-average BatchNorm (momentum=None)")
        meta["seg_off"] = [int(v) for v in seg_off]
        nseg = len(seg_off) - 1
    if bump:
        if batched_counters.active is not None:
            batched_counters.active.pending += [(t, nseg) for t in bump]
        else:
            torch._foreach_add_(bump, nseg)
    return meta


def group_hoist_ok(conv, bn, n_layers, B, N, S, K, D, device):
    """May a set-abstraction level run the feature share of its first conv on the source points?  (Not the first level: its few
    input features are no contraction worth moving, and in whole-tree execution its clouds are ragged.)"""
    return (D > 0 and n_layers >= 2 and device.type == "cuda" and GEMM_PRECISION == "f32" and bn is not None
            and (bn.training or bn.running_mean is not None) and conv.out_channels in HOIST_GROUP_WIDTHS and K <= 64 and S * K > N and D % 4 == 0 and D >= 32
            and B * S * K >= int(os.environ.get("PN2_HOIST_GROUP_MIN_ROWS", HOIST_GROUP_MIN_ROWS))
            and not os.environ.get("PN2_NO_HOIST") and not os.environ.get("PN2_NO_HOIST_GROUP")
            and not os.environ.get("PN2_NO_LAZY_ROWS"))


def group_bn_rows(xyz, new_xyz, feats, idx32, conv, bn, xyz_last=False, relu=True, seg_off=None):
    """The first layer of a set-abstraction level on its grouped rows without forming them: xyz [B,N,3], new_xyz [B,S,3],
    feats [B,N,D], idx32 [B,S,K] (ball query) -> LazyRows of relu(bn(conv([xyz[idx] - centre, feats[idx]]))) over the B*S*K
    rows (channel order [feats, xyz - centre] with xyz_last); see _GroupBNFn."""
    B, N, D = feats.shape
    W = conv.weight.reshape(conv.out_channels, -1)                      # [C, 3 + D] view of the parameter
    fcol, xcol = (0, D) if xyz_last else (3, 0)
    wf = W[:, fcol:fcol + D].contiguous()                               # (autograd sends its gradient back into the slice)
    bare = _BareConv(conv)
    bare.weight = wf
    gf = chain_rows(feats.reshape(B * N, D), [(bare, None, False)])     # [B*N, C], bias included
    meta = _bn_meta(bn, relu, seg_off, (bn.weight, bn.bias, conv.bias))
    meta["xcol"] = xcol
    y, st = _GroupBNFn.apply(gf.view(B, N, -1), xyz, new_xyz, idx32, W, bn.weight, bn.bias, conv.bias, meta)
    return LazyRows(y, st, bool(relu), meta.get("seg_off"), meta["lazy_handle"])


class _BareConv:
    """(weight, detached bias) of a conv module in the shape chain_rows expects of a layer's conv."""

    def __init__(self, conv):
        self.weight, self.out_channels = conv.weight, conv.out_channels
        self.bias = None if conv.bias is None else conv.bias.detach()


class _LinearAsConv:
    """An nn.Linear seen as the 1 x 1 conv it is (weight [C_out, C_in], bias), for chain_rows."""

    def __init__(self, lin):
        self.weight, self.bias, self.out_channels = lin.weight, lin.bias, lin.out_features


def linear_rows(rows, lin):
    """rows [R, C_in] -> lin(rows) [R, C_out] as a one-layer chain: the forward / input-gradient GEMM kernels of this library and the
    split-K weight gradient (deterministic slab sums) instead of the BLAS library's -- whose kernels for R ~ 10^6 rows and 32-512
    channels run at a tenth of the memory system's pace (a PointTransformerV3 training step spends 0.2 s in them, tools/prof_ptv3_torch.py)."""
    return chain_rows(rows, [(_LinearAsConv(lin), None, False)])


def hoisted_conv(rows, conv):
    """rows [R, C_in] -> conv(rows) [R, C_out], the bias applied but outside the graph (interp_bn_rows gives it its gradient)."""
    return chain_rows(rows, [(_BareConv(conv), None, False)])


def hoist_ok(conv, bn, n_layers, device):
    """May a feature-propagation level without a skip connection run its first conv in front of the interpolation?"""
    return (n_layers >= 2 and device.type == "cuda" and bn is not None and (bn.training or bn.running_mean is not None)
            and conv.out_channels in HOIST_WIDTHS and not os.environ.get("PN2_NO_HOIST")
            and not os.environ.get("PN2_NO_LAZY_ROWS"))


def _chain_spec(layers, pool_k, seg_off, dx_first_col):
    """(meta, params) of a chain call, or None for an empty chain; bumps num_batches_tracked."""
    layers = list(layers)
    if not layers:
        return None
    specs, params, training, bump = [], [], False, []
    for conv, bn, relu in layers:
        w = conv.weight.reshape(conv.out_channels, -1)
        spec = {"cout": conv.out_channels, "has_bn": bn is not None, "relu": bool(relu), "eps": 0.0, "momentum": 0.0,
                "running_mean": None, "running_var": None}
        if bn is not None:
            use_batch = bn.training or bn.running_mean is None
            training = training or use_batch
            spec["momentum"] = _bn_momentum(bn, bump)
            spec["eps"] = float(bn.eps)
            if bn.track_running_stats and bn.running_mean is not None:
                spec["running_mean"], spec["running_var"] = bn.running_mean, bn.running_var
            params += [w, conv.bias, bn.weight, bn.bias]
            spec["leaves"] = (conv.weight, conv.bias, bn.weight, bn.bias)
        else:
            params += [w, conv.bias, None, None]
            spec["leaves"] = (conv.weight, conv.bias, None, None)
        specs.append(spec)
    # a chain is either all batch statistics or all running statistics (module.train()/eval() sets them together)
    meta = {"layers": specs, "pool_k": int(pool_k), "training": training, "dx_first_col": int(dx_first_col)}
    nseg = 1
    if seg_off is not None and len(seg_off) > 2 and training:
        if any(bn is not None and bn.momentum is None for _, bn, _ in layers):
            raise NotImplementedError("row segments with cumulative-average BatchNorm (momentum=None)")
        if len(seg_off) - 1 > _hip.MAX_SEGMENTS:
            raise RuntimeError(f"chain_rows: at most {_hip.MAX_SEGMENTS} row segments per call, got {len(seg_off) - 1}")
        meta["seg_off"] = [int(v) for v in seg_off]
        nseg = len(seg_off) - 1
    if bump:
        inc = nseg if training else 1
        if batched_counters.active is not None:
            batched_counters.active.pending += [(t, inc) for t in bump]
        else:
            torch._foreach_add_(bump, inc)
    return meta, params


def _link(x, spec, seg_off):
    """x may be a LazyRows: -> (tensor to feed, meta with the link)."""
    if not isinstance(x, LazyRows):
        return x, spec[0]
    if (x.seg_off or None) != ([int(v) for v in seg_off] if seg_off is not None and len(seg_off) > 2 else None):
        raise RuntimeError("a linked chain must use the row segments of the chain that produced its input")
    if x.consumed and not x.handle.get("shared"):
        raise RuntimeError("a LazyRows feeds ONE consumer (a chain, or the two chains of chain_pair_rows)")
    meta = dict(spec[0])
    meta["link"] = x
    return x.y, meta


def chain_rows(x, layers, pool_k=1, seg_off=None, dx_first_col=0, lazy_out=False):
    """x [R, C_in] fp32 rows; layers: iterable of (conv, bn_or_None, relu: bool).
    -> [R, C_out], or [R // pool_k, C_out] (max over each group of pool_k consecutive rows) when pool_k > 1.
    seg_off: optional ascending row offsets [0, ..., R] of the mini-batches the rows are made of (whole-tree execution):
    train-mode BatchNorm then works per segment, exactly as if the segments had been separate calls, and every layer's
    running statistics / num_batches_tracked advance once per segment.
    dx_first_col: the gradient w.r.t. x is only needed from this column on (the leading columns come back as zeros).
    lazy_out: return a LazyRows (the final BatchNorm + ReLU left to the consumer, a linked chain) when the chain ends in a
    train-mode BatchNorm on the device; x itself may be a LazyRows."""
    spec = _chain_spec(layers, pool_k, seg_off, dx_first_col)
    if spec is None:
        return x
    xin, meta = _link(x, spec, seg_off)
    last = meta["layers"][-1]
    if lazy_out and last["has_bn"] and meta["training"] and pool_k == 1 and xin.is_cuda and not os.environ.get("PN2_NO_LAZY_ROWS"):
        meta = dict(meta)
        meta["lazy_out"] = True
        meta["lazy_handle"] = {}
        y, st = _ChainFn.apply(xin, meta, *spec[1])
        return LazyRows(y, st, last["relu"], meta.get("seg_off"), meta["lazy_handle"])
    if isinstance(x, LazyRows):
        x.consumed = True
    return _ChainFn.apply(xin, meta, *spec[1])


def chain_pair_rows(x, layers_a, layers_b, seg_off=None):
    """Two chains reading the same rows x [R, C_in] (the two prediction heads on the backbone features) ->
    (out_a, out_b); one autograd node, the input gradient leaves as ONE tensor (see _ChainPairFn).  Falls back to two
    chain_rows calls when either chain is empty or starts with a narrow layer."""
    layers_a, layers_b = list(layers_a), list(layers_b)
    dev_ok = x.y.is_cuda if isinstance(x, LazyRows) else x.is_cuda
    if len(layers_a) < 2 or len(layers_b) < 2 or not dev_ok or os.environ.get("PN2_NO_CHAIN_PAIR"):
        if isinstance(x, LazyRows):
            x.handle["shared"] = True                   # two separate consumers: neither sees the complete gradient
        return chain_rows(x, layers_a, seg_off=seg_off), chain_rows(x, layers_b, seg_off=seg_off)
    sa, sb = _chain_spec(layers_a, 1, seg_off, 0), _chain_spec(layers_b, 1, seg_off, 0)
    xin, ma = _link(x, sa, seg_off)
    _, mb = _link(x, sb, seg_off)
    if isinstance(x, LazyRows):
        x.consumed = True
    return _ChainPairFn.apply(xin, ma, mb, len(sa[1]), *sa[1], *sb[1])
