"""CPU: the torch-CPU restatement (oracle/torch_port.py, the benchmark's cpu_baseline) against the golden
vectors of the imported reference.  Runs the same BLAS as the fixture generator, so indices are bit-exact
here; float outputs are compared at fp32 rounding level."""
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import torch_port as P


def gold(name):
    return np.load(os.path.join(helpers.GOLDEN, name))


def test_port_ops_match_golden():
    g = gold("ops.npz")
    xyz = torch.from_numpy(g["coords"]).permute(0, 2, 1)
    fps = P.farthest_point_sample(xyz, 128, torch.from_numpy(g["fps_start"]))
    assert np.array_equal(fps.numpy(), g["fps_idx"])
    new_xyz = P.index_points(xyz, fps)
    assert np.array_equal(new_xyz.numpy(), g["new_xyz"])
    for tag, r, K in [("r01", 0.1, 32), ("r02", 0.2, 32), ("r005", 0.05, 16)]:
        assert np.array_equal(P.query_ball_point(r, K, xyz, new_xyz).numpy(), g[f"bq_{tag}"])
    assert np.array_equal(P.query_ball_point(0.1, 32, xyz, torch.from_numpy(g["q_shift"])).numpy(), g["bq_shift"])
    interp = P.three_nn_interpolate(xyz, new_xyz, torch.from_numpy(g["interp_points2"]))
    assert np.array_equal(interp.permute(0, 2, 1).numpy(), g["interp_out"])


@pytest.mark.parametrize("depth", [5, 4, 6])
def test_port_model_matches_golden(depth):
    g = gold(f"model_d{depth}.npz")
    torch.manual_seed(int(g["weight_seed"]))
    model = P.PortPointNet2(depth=depth)
    ps = sorted(model.named_parameters(), key=lambda kv: kv[0])
    assert [n for n, _ in ps] == [str(n) for n in g["grad_names"]]
    np.testing.assert_allclose([float(p.detach().double().sum()) for _, p in ps], g["param_sum"], rtol=0, atol=1e-9)
    model.train()
    batch = {k: torch.from_numpy(g[k]) for k in ["coords", "feats", "masks_pad", "masks_off", "semantic_labels",
                                                  "offset_labels"]}
    torch.manual_seed(int(g["torch_seed"]))
    loss, ld, sem, off = P.loss_from_batch(model, batch)
    (loss * 50).backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    scale = float(np.abs(g["offset_predictions"]).max())
    assert float(np.abs(off.detach().numpy() - g["offset_predictions"]).max()) <= 1e-5 * scale
    params = dict(model.named_parameters())
    gmax = float(g["grad_l2"].max())
    for name, l2 in zip(g["grad_names"], g["grad_l2"]):
        name = str(name)
        got = float(params[name].grad.double().norm())
        if helpers.is_pre_bn_bias(name):
            wn = float(params[name[:-4] + "weight"].grad.double().norm())
            assert got <= 1e-2 * wn and l2 <= 1e-2 * wn, name      # both are rounding noise around zero
        else:
            assert abs(got - l2) <= 2e-4 * l2 + 1e-6 * gmax, name
