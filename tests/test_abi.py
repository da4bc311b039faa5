"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pn2_hip.h declares.
No compute call is made (there is no GPU here); argument validation paths that return before any HIP call are
exercised because they are pure host code."""
import ctypes
import os
import re

import pytest

import helpers

pn2 = helpers.load_pkg()


@pytest.fixture(scope="module")
def cdll():
    from pn2_amd import build, _hip
    build.build()
    return ctypes.CDLL(_hip.LIB_PATH)


def declared_symbols():
    src = open(os.path.join(helpers.REPO, "include", "pn2_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(cdll):
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in pn2_hip.h but not exported"


def test_binding_table_matches_header():
    from pn2_amd import _hip
    assert sorted(_hip.SIGNATURES) == declared_symbols()


def test_version_and_arch(cdll):
    from pn2_amd import _hip
    cdll.pn2_arch.restype = ctypes.c_char_p
    assert cdll.pn2_version() == _hip.ABI_VERSION == 6
    assert cdll.pn2_arch() == b"gfx950"


def test_code_object_is_gfx950_only():
    from pn2_amd import _hip
    blob = open(_hip.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"sm_80"):
        assert other not in blob


def test_bad_arguments_are_rejected_on_the_host(cdll):
    cdll.pn2_fps_workspace_bytes.restype = ctypes.c_size_t
    assert cdll.pn2_fps_workspace_bytes(0, 10, 4) == 0
    assert cdll.pn2_fps_workspace_bytes(1, 2_000_000, 4) == 0          # beyond 64 members x 16384 points
    assert cdll.pn2_fps_workspace_bytes(1, 1024, 16) == 16              # single workgroup: error word only
    # XCD-local ordered multi-pick rounds, 32 members x 16 points per lane: header + 9 granules {2 x (key, x, y, z), bound} per
    # member and round (at most npoint rounds), then -- each 256-byte aligned -- the cloud's box, 2 x 4096 cell counters, the
    # 4097 cell starts and the cell order of the points
    up = lambda v: (v + 255) // 256 * 256
    box = up(64 + 1024 * 9 * 32 * 8)
    order = up(up(up(box + 8 * 4) + 2 * 4096 * 4) + 4097 * 4)
    assert cdll.pn2_fps_workspace_bytes(1, 262144, 1024) == up(order + 262144 * 4) + 3 * 262144 * 4     # + coordinates in cell order
    order1 = order
    box = up(64 + 8 * 1024 * 9 * 32 * 8)                                 # 8 clouds x 32 members x 4 points per lane
    order = up(up(up(box + 8 * 8 * 4) + 8 * 2 * 4096 * 4) + 8 * 4097 * 4)
    assert cdll.pn2_fps_workspace_bytes(8, 65536, 1024) == up(order + 8 * 65536 * 4) + 3 * 8 * 65536 * 4
    null = ctypes.c_void_p(None)
    i64 = ctypes.c_int64
    # null pointers -> PN2_E_BADARG before any HIP call
    assert cdll.pn2_fps_f32(null, i64(0), i64(0), i64(0), 1, 8, 4, null, null, null, null, ctypes.c_size_t(0), null,
                            null) == -1
    assert cdll.pn2_three_nn_f32(null, i64(0), i64(0), i64(0), null, i64(0), i64(0), i64(0), 1, 8, 3, null, null, null,
                                 null, null) == -1
    cdll.pn2_fps_order_offset.restype = ctypes.c_size_t
    assert cdll.pn2_fps_order_offset(1, 1024, 16) == ctypes.c_size_t(-1).value      # no cell order for a single workgroup
    assert cdll.pn2_fps_order_offset(1, 262144, 1024) == order1


def test_cpu_tensors_fail_loudly():
    import torch
    from pn2_amd.PointNet2 import pointnet2_utils as U
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        U.query_ball_point(0.1, 8, torch.zeros(1, 16, 3), torch.zeros(1, 4, 3))


def test_product_never_imports_oracle():
    for root, _, files in os.walk(helpers.PKG_DIR):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", text, flags=re.M), f"{f} imports the oracle"
                assert "libpn2oracle" not in text and "pn2o_" not in text, f"{f} binds the oracle library"
