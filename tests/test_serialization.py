"""Point-cloud serialization (first stage of PointTransformerV3): space-filling-curve codes.
CPU: the numpy oracle against the fixture generated from the reference's own module (tests/golden/make_serialization.py) --
integer work, bit for bit.  GPU: csrc/serialize.hip against the fixture and the oracle, encode -> decode round trips at
BASELINE configs[3]'s 1 048 576 points, and the code / order / inverse triple of Point.serialization."""
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import serialization_port as S

GOLD = os.path.join(os.path.dirname(__file__), "golden", "serialization.npz")
DEPTHS = (16, 10, 8, 5, 1)


@pytest.fixture(scope="module")
def pn2():
    return helpers.load_pkg()


def test_oracle_equals_the_reference_fixture():
    g = np.load(GOLD)
    for depth in DEPTHS:
        grid, batch = g[f"grid_d{depth}"], g[f"batch_d{depth}"]
        for order in S.ORDERS:
            assert np.array_equal(S.encode(grid, batch, depth, order), g[f"code_d{depth}_{order}"]), (depth, order)
            assert np.array_equal(S.encode(grid, None, depth, order), g[f"code_nobatch_d{depth}_{order}"]), (depth, order)
        for order in ("z", "hilbert"):
            dg, db = S.decode(g[f"code_d{depth}_{order}"], depth, order)
            assert np.array_equal(dg, g[f"decoded_grid_d{depth}_{order}"]) and np.array_equal(db, g[f"decoded_batch_d{depth}_{order}"])
            assert np.array_equal(dg, grid.astype(np.int64))          # decode inverts encode


def test_oracle_curve_properties():
    """A Hilbert curve visits every cell of the cube exactly once and consecutive codes are face neighbours; z-order is a
    bijection too."""
    depth = 4
    cells = np.stack(np.meshgrid(*[np.arange(1 << depth)] * 3, indexing="ij"), -1).reshape(-1, 3)
    for order in S.ORDERS:
        code = S.encode(cells, None, depth, order)
        assert np.array_equal(np.sort(code), np.arange(1 << 3 * depth))
    walk = cells[np.argsort(S.encode(cells, None, depth, "hilbert"))]
    assert np.all(np.abs(np.diff(walk, axis=0)).sum(1) == 1)


@pytest.mark.gpu
def test_hip_codes_equal_the_reference_fixture(pn2):
    from pn2_amd.PointTransformerV3 import serialization as ser
    g = np.load(GOLD)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    for depth in DEPTHS:
        grid, batch = dev(g[f"grid_d{depth}"]), dev(g[f"batch_d{depth}"])
        for order in S.ORDERS:
            assert np.array_equal(ser.encode(grid, batch, depth, order).cpu().numpy(), g[f"code_d{depth}_{order}"]), (depth, order)
            assert np.array_equal(ser.encode(grid.long(), None, depth, order).cpu().numpy(), g[f"code_nobatch_d{depth}_{order}"])
        for order in ("z", "hilbert"):
            dg, db = ser.decode(dev(g[f"code_d{depth}_{order}"]), depth, order)
            assert np.array_equal(dg.cpu().numpy(), g[f"decoded_grid_d{depth}_{order}"])
            assert np.array_equal(db.cpu().numpy(), g[f"decoded_batch_d{depth}_{order}"])


@pytest.mark.gpu
def test_hip_serialization_at_full_size(pn2):
    """1 048 576 voxels (BASELINE configs[3]): all four orders from one launch equal the oracle bit for bit; encode -> decode
    is the identity; order / inverse are inverse permutations that sort the codes; strided and negative inputs."""
    from pn2_amd.PointTransformerV3 import serialization as ser
    rng = np.random.default_rng(3)
    n, depth = 1 << 20, 16
    grid = rng.integers(0, 1 << depth, size=(n, 3)).astype(np.int32)
    batch = np.sort(rng.integers(0, 4, size=n)).astype(np.int64)
    tg, tb = torch.as_tensor(grid, device="cuda"), torch.as_tensor(batch, device="cuda")
    code, order, inverse = ser.serialize(tg, tb, depth, list(S.ORDERS))
    for k, o in enumerate(S.ORDERS):
        assert np.array_equal(code[k].cpu().numpy(), S.encode(grid, batch, depth, o)), o
    assert torch.equal(torch.gather(inverse, 1, order), torch.arange(n, device="cuda").expand(4, n))
    assert bool((torch.diff(torch.gather(code, 1, order), dim=1) >= 0).all())
    for o, k in (("z", 0), ("hilbert", 2)):
        dg, db = ser.decode(code[k], depth, o)
        assert torch.equal(dg, tg.long()) and torch.equal(db, tb)
    # a strided view, and coordinates with bits above `depth` / negative values: only the low bits count (like .long() & mask)
    wide = torch.as_tensor(rng.integers(-(1 << 20), 1 << 20, size=(5000, 6)).astype(np.int32), device="cuda")
    view = wide[:, ::2]
    for o in S.ORDERS:
        assert np.array_equal(ser.encode(view, None, 10, o).cpu().numpy(), S.encode(view.cpu().numpy(), None, 10, o))
    with pytest.raises(RuntimeError):
        ser.encode(tg, None, 17, "z")
    with pytest.raises(RuntimeError):
        ser.encode(tg.cpu(), None, 16, "z")
