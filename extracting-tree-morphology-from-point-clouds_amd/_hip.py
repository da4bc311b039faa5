"""ctypes binding of libpn2hip.so (C ABI: include/pn2_hip.h).

The library is loaded on first use and the import fails LOUDLY when it is missing: there is no CPU fallback
and nothing here ever routes through oracle/.  Tensors only provide device memory and the stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PN2_LIB") or os.path.join(_HERE, "libpn2hip.so")   # (PN2_LIB: a diagnostic build, tools/diag_*.py)

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int
_f32 = ctypes.c_float
_sz = ctypes.c_size_t



class MLPLayer(ctypes.Structure):
    """struct pn2_mlp_layer of include/pn2_hip.h (field order and types must match)."""
    _fields_ = [("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("weight", _vp), ("bias", _vp),
                ("has_bn", ctypes.c_int32), ("relu", ctypes.c_int32), ("gamma", _vp), ("beta", _vp),
                ("running_mean", _vp), ("running_var", _vp), ("eps", _f32), ("momentum", _f32),
                ("y", _vp), ("stats", _vp), ("dweight", _vp), ("dbias", _vp), ("dgamma", _vp), ("dbeta", _vp),
                ("in_stats", _vp), ("in_relu", ctypes.c_int32), ("in_partial", _vp), ("out_partial", _vp),
                ("out_partial_rows", ctypes.c_int32), ("out_partial_cpb", ctypes.c_int32)]


_lp = ctypes.POINTER(MLPLayer)
WGRAD_TASKS_MAX = 64                 # PN2_WGRAD_TASKS_MAX


class WgradTask(ctypes.Structure):
    """struct pn2_wgrad_task: one deferred split-K slab reduction (slab [nsplit][mn] -> out [mn] +=)."""
    _fields_ = [("slab", _vp), ("out", _vp), ("mn", ctypes.c_int64), ("nsplit", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class WgradTasks(ctypes.Structure):
    """struct pn2_wgrad_tasks: the HOST list a chain backward appends its deferred reductions to."""
    _fields_ = [("n", ctypes.c_int32), ("reserved", ctypes.c_int32), ("t", WgradTask * WGRAD_TASKS_MAX)]


_tp = ctypes.POINTER(WgradTasks)


class Coop(ctypes.Structure):
    """struct pn2_coop: cooperative chain launches (sync: >= 64 zeroed device uint32, one buffer per stream; status word)."""
    _fields_ = [("sync", _vp), ("status", _vp), ("spin_limit", ctypes.c_uint32), ("max_workgroups", ctypes.c_int32)]


_cp = ctypes.POINTER(Coop)


class Segments(ctypes.Structure):
    """struct pn2_segments of include/pn2_hip.h: row_off is a HOST int32 array of nseg + 1 offsets."""
    _fields_ = [("nseg", ctypes.c_int32), ("row_off", ctypes.POINTER(ctypes.c_int32))]


_sp = ctypes.POINTER(Segments)
MAX_SEGMENTS = 128                   # PN2_MAX_SEGMENTS


def segments_arg(row_off):
    """row_off: sequence of nseg + 1 ascending ints (or None) -> (ctypes pointer or None, keep-alive object)."""
    if row_off is None or len(row_off) <= 2:
        return None, None
    arr = (ctypes.c_int32 * len(row_off))(*[int(v) for v in row_off])
    seg = Segments(len(row_off) - 1, ctypes.cast(arr, ctypes.POINTER(ctypes.c_int32)))
    return ctypes.pointer(seg), (arr, seg)

# name -> (restype, argtypes); must list every symbol include/pn2_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "pn2_version": (_int, []),
    "pn2_mask_ranks": (_int, [_vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "pn2_point_loss_workspace_bytes": (_sz, [_int]),
    "pn2_point_loss_fwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _int, _vp, _vp, _sz, _vp]),
    "pn2_point_loss_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "pn2_point_loss_weighted_fwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp, _sz,
                                               _vp]),
    "pn2_point_loss_weighted_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _int, _vp, _vp, _vp, _vp,
                                               _vp]),
    "pn2_knn_radius_f64": (_int, [_vp, _int, _int, ctypes.c_double, _vp, _vp, _vp, _vp]),
    "pn2_cov_eig_f64": (_int, [_vp, _int, _vp, _int, _int, _vp, _vp, _vp]),
    "pn2_knn_grid_workspace_bytes": (_sz, [_int]),
    "pn2_knn_radius_grid_f64": (_int, [_vp, _int, _int, ctypes.c_double, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pn2_arch": (ctypes.c_char_p, []),
    "pn2_square_distance_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _int, _int, _int, _vp, _vp]),
    "pn2_fps_workspace_bytes": (_sz, [_int, _int, _int]),
    "pn2_fps_order_offset": (_sz, [_int, _int, _int]),
    "pn2_fps_rounds_offset": (_sz, [_int, _int, _int]),
    "pn2_fps_box_offset": (_sz, [_int, _int, _int]),
    "pn2_fps_cellstart_offset": (_sz, [_int, _int, _int]),
    "pn2_fps_sorted_xyz_offset": (_sz, [_int, _int, _int]),
    "pn2_fps_f32": (_int, [_vp, _i64, _i64, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "pn2_ball_query_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "pn2_ball_query_cells_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _int, _int, _int, _f32, _int, _vp, _vp, _vp,
                                        _vp, _vp, _vp]),
    "pn2_ball_query_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _int, _int, _int, _f32, _int, _vp, _vp,
                                  _sz, _vp]),
    "pn2_group_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _int, _int,
                             _vp, _vp, _vp]),
    "pn2_group_grad_f32": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _vp]),
    "pn2_gather_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _vp, _vp, _vp]),
    "pn2_gather_grad_f32": (_int, [_vp, _vp, _int, _int, _int, _int, _vp, _vp]),
    "pn2_three_nn_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "pn2_three_interpolate_f32": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _i64, _vp, _vp]),
    "pn2_three_interpolate_concat_f32": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _i64, _vp, _vp]),
    "pn2_three_interpolate_grad_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "pn2_three_interpolate_grad_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "pn2_serialize_encode_i64": (_int, [_vp, _i64, _i64, _vp, ctypes.c_longlong, _int, ctypes.POINTER(ctypes.c_int32), _int, _vp,
                                        _vp]),
    "pn2_serialize_decode_i64": (_int, [_vp, ctypes.c_longlong, _int, _int, _vp, _vp, _vp]),
    "pn2_ptv3_subm_workspace_bytes": (_sz, [_int]),
    "pn2_ptv3_subm_neighbors_i32": (_int, [_vp, _vp, _int, _int, _vp, _vp, _sz, _vp, _vp]),
    "pn2_ptv3_subm_conv_f32": (_int, [_vp, _i64, _vp, _int, _vp, _vp, _vp, _int, _int, _int, _vp, _i64, _vp]),
    "pn2_ptv3_pad_unpad_i64": (_int, [_vp, _vp, _vp, _int, _int, _i64, _vp, _vp, _vp, _vp]),
    "pn2_ptv3_patch_attention_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, _f32, _vp, _int, _vp]),
    "pn2_ptv3_patch_attention_lse_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, _f32, _vp, _vp, _int, _vp]),
    "pn2_ptv3_patch_attention_bwd_f32": (_int, [_vp, _i64, _vp, _i64, _int, _int, _int, _f32, _vp, _vp, _vp, _vp, _vp]),
    "pn2_ptv3_subm_wgrad_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "pn2_layer_norm_supported": (_int, [_int]),
    "pn2_layer_norm_fwd_f32": (_int, [_vp, _i64, _vp, _vp, _f32, _i64, _int, _vp, _i64, _vp, _vp, _vp]),
    "pn2_layer_norm_bwd_workspace_bytes": (_sz, [_i64, _int]),
    "pn2_layer_norm_bwd_f32": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _int, _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "pn2_ptv3_subm_wgrad_f32": (_int, [_vp, _i64, _vp, _int, _vp, _i64, _int, _int, _int, _vp, _vp, _sz, _vp]),
    "pn2_prof_enable": (None, [_int]),
    "pn2_prof_collect": (_int, [ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong),
                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _int]),
    "pn2_mlp_workspace_bytes": (_sz, [_int, _lp, _int, _int]),
    "pn2_mlp_chain_bf16_storage": (_int, [_int, _lp, _int, _int]),
    "pn2_mlp_chain_fwd_f32": (_int, [_vp, _i64, _int, _lp, _int, _int, _int, _vp, _vp, _sp, _int, _cp, _vp, _sz, _vp]),
    "pn2_mlp_link_partial_bytes": (_sz, [_int, _int, _int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "pn2_mlp_reduce_wgrad": (_int, [ctypes.POINTER(WgradTask), _int, _vp]),
    "pn2_mlp_pair_dgrad_f32": (_int, [_int, _lp, _vp, _lp, _vp, _vp, _i64, _vp, _i64, _sp, _int, _vp]),
    "pn2_group_bn_workspace_bytes": (_sz, [_int, _int, _int, _int, _int]),
    "pn2_group_bn_fwd_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _lp, _sp, _int, _vp, _vp,
                                    _sz, _vp]),
    "pn2_group_bn_bwd_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _int, _int, _int, _lp, _vp, _vp, _i64, _sp, _vp, _sz,
                                    _vp]),
    "pn2_interp_bn_workspace_bytes": (_sz, [_int, ctypes.c_longlong, _int, _int, _int]),
    "pn2_interp_bn_fwd_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, ctypes.c_longlong, _lp, _sp, _int, _int, _vp, _vp,
                                     _sz, _vp]),
    "pn2_interp_bn_bwd_f32": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, ctypes.c_longlong, _lp, _vp, _sp, _int, _vp, _sz, _vp]),
    "pn2_mlp_chain_bwd_f32": (_int, [_vp, _i64, _int, _lp, _int, _int, _vp, _vp, _vp, _i64, _int, _vp, _vp, _sp, _int, _tp, _cp, _vp,
                                     _sz, _vp]),
    "pn2_cylinder_project_f32": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "pn2_raster_ranges_f32": (_int, [_vp, _i64, _int, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "pn2_raster_keys": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp]),
    "pn2_raster_pack_f32": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "pn2_fps_ragged_workspace_bytes": (_sz, [_int, _int, _int]),
    "pn2_fps_ragged_f32": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pn2_ball_query_ragged_f32": (_int, [_vp, _vp, _vp, _int, _int, _int, _f32, _int, _vp, _vp]),
    "pn2_group_ragged_f32": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp]),
    "pn2_three_nn_ragged_f32": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp, _vp, _vp]),
    "pn2_three_interpolate_ragged_f32": (_int, [_vp, _vp, _vp, _vp, _int, _int, ctypes.c_longlong, _int, _int, _vp, _i64, _i64,
                                                _vp, _vp]),
    "pn2_three_interpolate_grad_ragged_workspace_bytes": (_sz, [_int, ctypes.c_longlong, _int]),
    "pn2_three_interpolate_grad_ragged_f32": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _int, _int, ctypes.c_longlong, _int, _int,
                                                     _vp, _vp, _sz, _vp]),
}

_lib = None
ABI_VERSION = 6                      # PN2_ABI_VERSION of include/pn2_hip.h
CHAIN_ACCUMULATE_DX = 0x100          # PN2_CHAIN_ACCUMULATE_DX
CHAIN_LAZY_OUT = 0x400               # PN2_CHAIN_LAZY_OUT
CHAIN_ZERO_LEAD = 0x800              # PN2_CHAIN_ZERO_LEAD
CHAIN_X_BF16, CHAIN_STORE_BF16, CHAIN_DOUT_BF16, CHAIN_DX_BF16 = 0x1000, 0x2000, 0x4000, 0x8000   # PN2_CHAIN_*_BF16
STATUS_FPS_HANDOFF, STATUS_FPS_ARRIVAL, STATUS_BAD_INDEX, STATUS_COOP_BARRIER = 1, 2, 4, 8   # PN2_STATUS_* bits


def lib():
    """Load (once) and return the bound library; raises if the HIP extension was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python build.py` (hipcc --offload-arch=gfx950). "
                "pn2_amd has no CPU fallback.")
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if cdll.pn2_version() != ABI_VERSION:
            raise RuntimeError(f"libpn2hip ABI {cdll.pn2_version()} != {ABI_VERSION}")
        _lib = cdll
    return _lib


def check(status, what):
    if status != 0:
        kind = "bad arguments" if status == -1 else "workspace too small" if status == -2 else f"hipError {status}"
        raise RuntimeError(f"libpn2hip: {what} failed ({kind})")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pn2_amd ops run on a HIP device only (no CPU fallback); got a CPU tensor")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current HIP stream of the current device as an integer handle.  torch.cuda.current_stream() builds a Stream object per
    call (8 us, 44 calls per headline step = 0.35 ms of a 3.6 ms host path); the raw accessor torch's own extensions use is ~0.3 us."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def f32(t):
    return t if t.dtype == torch.float32 else t.float()


class KernelTimer:
    """Optional per-entry-point timing with HIP events on the launch stream (used by bench.py for the roofline
    object; off by default and then costs one attribute test per call)."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (name, algorithmic_bytes, flops, start_event, end_event)

    def reset(self):
        self.records = []

    def summary(self):
        """name -> dict(calls, ms, bytes, flops); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, nbytes, flops, e0, e1 in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
            d["flops"] += flops
        return out


timer = KernelTimer()


def call(name, fn, *args, nbytes=0, flops=0):
    """Invoke one C-ABI entry point and raise on a non-zero status; `nbytes`/`flops` are the ALGORITHMIC traffic
    and work of this launch, recorded only when the timer is on."""
    if timer.enabled:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        status = fn(*args)
        e1.record()
        timer.records.append((name, int(nbytes), int(flops), e0, e1))
    else:
        status = fn(*args)
    check(status, name)


def kernel_profile(fn):
    """Run fn() with the library's per-kernel HIP-event timing on; returns a list of dicts
    (name, calls, ms, bytes, flops) where bytes/flops are the ALGORITHMIC figures of ONE launch of that group."""
    L = lib()
    L.pn2_prof_enable(1)
    try:
        fn()
    finally:
        L.pn2_prof_enable(0)
    cap, maxg = 1 << 16, 1024
    names = ctypes.create_string_buffer(cap)
    ms = (ctypes.c_double * maxg)()
    calls = (ctypes.c_longlong * maxg)()
    nbytes = (ctypes.c_double * maxg)()
    flops = (ctypes.c_double * maxg)()
    n = L.pn2_prof_collect(names, cap, ms, calls, nbytes, flops, maxg)
    out, raw = [], names.raw
    pos = 0
    for i in range(n):
        end = raw.index(b"\0", pos)
        out.append({"name": raw[pos:end].decode(), "calls": int(calls[i]), "ms": float(ms[i]), "bytes": float(nbytes[i]),
                    "flops": float(flops[i])})
        pos = end + 1
    return out
