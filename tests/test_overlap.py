"""Overlapping rasters (stride < size: the reference's training default, train_PointNet2.py:84-85,109).  A point id then occurs
more than once inside one mini-batch and the reference's `avg[point_ids] += x; count[point_ids] += 1` (PointNet2.py:272-276,
376-380) is an index_put WITHOUT accumulation: one duplicate per mini-batch lands (the last one on the CPU), count + 1.

CPU half: streaming.last_occurrence / RefScatter / RefPut against the reference's literal expression run by torch on the CPU
(values, counts and the gradients autograd derives for it).  GPU half: forward_hierarchical_streaming and forward_hierarchical,
both execution modes, against fixtures produced by the REFERENCE's own methods on overlapping rasters
(tests/golden/make_golden.py --only-overlap, make_golden_hier.py --overlap): losses, gradients and returned predictions."""
import numpy as np
import pytest
import torch

import helpers
from test_hip_parity import _FakeScaler, close_to_reference, dev, gold  # noqa: F401


def _stream(seed, n=50, M=4, rows=40):
    g = torch.Generator().manual_seed(seed)
    ids = [torch.randint(0, n, (rows,), generator=g) for _ in range(M)]
    vals = [torch.randn(rows, 3, generator=g, dtype=torch.float64) for _ in range(M)]
    moff = [torch.rand(rows, generator=g) > 0.3 for _ in range(M)]
    return ids, vals, moff


def _reference_expression(ids, vals, moff, n, w):
    """the reference's loop, literally, on the CPU (autograd included)"""
    vals = [v.clone().requires_grad_(True) for v in vals]
    total = torch.zeros(n, 3, dtype=torch.float64)
    count = torch.zeros(n, 1, dtype=torch.float64)
    for i, v, m in zip(ids, vals, moff):
        total[i[m]] += v[m]
        count[i[m]] += 1
    seen = count.squeeze(1) > 0
    total[seen] /= count[seen]
    (total * w).sum().backward()
    return total.detach(), count, [v.grad for v in vals]


def test_reference_overlap_semantics_on_cpu():
    helpers.load_pkg()
    from pn2_amd import streaming
    n, M = 50, 4
    ids, vals, moff = _stream(3, n, M)
    assert any(len(torch.unique(i)) < len(i) for i in ids)
    w = torch.randn(n, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(9))
    want, want_count, want_grads = _reference_expression(ids, vals, moff, n, w)

    # fused form: all mini-batches at once
    acc = streaming._Accumulators(n, "cpu")
    acc.off, acc.sem = acc.off.double(), acc.sem.double()
    v = [x.clone().requires_grad_(True) for x in vals]
    seg = torch.cat([torch.full((len(i),), j, dtype=torch.long) for j, i in enumerate(ids)])
    acc.add(torch.cat(ids), torch.zeros(sum(len(i) for i in ids), 2, dtype=torch.float64), torch.cat(v), torch.cat(moff),
            differentiable=True, seg=seg, M=M)
    out = acc.average()["offset_predictions"]
    assert torch.equal(acc.off_cnt.double(), want_count)
    torch.testing.assert_close(out, want, rtol=1e-12, atol=1e-12)
    (out * w).sum().backward()
    for a, b in zip(v, want_grads):
        torch.testing.assert_close(a.grad, b, rtol=1e-12, atol=1e-12)

    # mini-batch by mini-batch form
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    total, count = torch.zeros(n, 3, dtype=torch.float64), torch.zeros(n, 1, dtype=torch.float64)
    v = [x.clone().requires_grad_(True) for x in vals]
    for i, x, m in zip(ids, v, moff):
        total = PointNet2._scatter_minibatch(total, count, i[m], x[m], differentiable=True)
    total = PointNet2._average(total, count)
    assert torch.equal(count, want_count)
    torch.testing.assert_close(total, want, rtol=1e-12, atol=1e-12)
    (total * w).sum().backward()
    for a, b in zip(v, want_grads):
        torch.testing.assert_close(a.grad, b, rtol=1e-12, atol=1e-12)


def test_average_mode_is_an_opt_in(monkeypatch):
    helpers.load_pkg()
    from pn2_amd import streaming
    ids, vals, moff = _stream(4, 30, 2)
    out = {}
    for mode in ("reference", "average"):
        monkeypatch.setenv("PN2_OVERLAP", mode)
        acc = streaming._Accumulators(30, "cpu")
        seg = torch.cat([torch.full((len(i),), j, dtype=torch.long) for j, i in enumerate(ids)])
        acc.add(torch.cat(ids), torch.zeros(80, 2), torch.cat(vals).float(), torch.cat(moff), differentiable=False, seg=seg, M=2)
        out[mode] = acc.average()["offset_predictions"]
    assert not torch.equal(out["reference"], out["average"])
    monkeypatch.setenv("PN2_OVERLAP", "nonsense")
    with pytest.raises(ValueError):
        streaming.overlap_mode()


# ------------------------------------------------------------------------------------------------------------ GPU
def _overlap_minibatches(g):
    from pn2_amd.synthetic import gaussian_branch_tree, rasterize
    xyz, off, _ = gaussian_branch_tree(20000, seed=5)
    n = len(xyz)
    feats_all = (np.sin(0.61 * np.arange(n * 4, dtype=np.float64) + 7)).astype(np.float32).reshape(n, 4)
    rasters = [g[f"raster{i}"].astype(np.int64) for i in range(int(g["n_rasters"]))]
    assert all(np.array_equal(a, b) for a, b in zip(rasters, [r for r in rasterize(xyz, 2.0, 1.0) if len(r) >= 40][:6]))
    mbs = []
    for k in range(0, len(rasters), 2):
        group = rasters[k:k + 2]
        nmax = max(len(r) for r in group)
        coords = np.zeros((len(group), 3, nmax), np.float32)
        fts = np.zeros((len(group), 4, nmax), np.float32)
        mpad = np.zeros((len(group), nmax), bool)
        for i, r in enumerate(group):
            coords[i, :, :len(r)] = (xyz[r] - np.floor(xyz[r].min(axis=0))).T
            fts[i, :, :len(r)] = feats_all[r].T
            mpad[i, :len(r)] = True
        ids = np.concatenate(group)
        assert len(np.unique(ids)) < len(ids)                # duplicates INSIDE the mini-batch: the case under test
        mbs.append({"coords": dev(coords), "feats": dev(fts), "masks_pad": dev(mpad), "masks_off": dev((np.arange(len(ids)) % 5) != 2),
                    "point_ids": dev(ids)})
    labels = {"cloud_length": n, "semantic_labels": torch.from_numpy((np.arange(n) % 3 == 0).astype(np.int64))[:, None],
              "offset_labels": torch.from_numpy(off)}
    return mbs, labels, rasters, n


def _check_grads(model, g, what):
    params = dict(model.named_parameters())
    gmax = float(g["grad_l2_f64"].max())
    noise = max(abs(l2 - l64) / l64 for n_, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"])
                if not helpers.is_pre_bn_bias(str(n_)) and l64 > 1e-3 * gmax)
    print(f"{what}: reference fp32 gradient-norm noise level {noise:.2e}")
    for name, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"]):
        name = str(name)
        if helpers.is_pre_bn_bias(name):
            continue
        got = float(params[name].grad.double().norm())
        bar = max(5e-4 * l64, 2 * noise * l64, 3 * abs(l2 - l64)) + 1e-6 * gmax
        assert abs(got - l64) <= bar, f"{what}: grad norm of {name}: {got} vs {l64}"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fused", "sequential"])
def test_streaming_overlapping_rasters_golden(mode, monkeypatch):
    """forward_hierarchical_streaming on rasters of size 2 / stride 1 against the reference's own loop (CPU): losses,
    accumulated gradients and the RETURNED predictions (one contribution per id and mini-batch -- the last duplicate)."""
    helpers.load_pkg()
    monkeypatch.setenv("PN2_STREAMING", mode)
    monkeypatch.delenv("PN2_OVERLAP", raising=False)
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    g = gold("streaming_overlap_d5.npz")
    mbs, labels, rasters, n = _overlap_minibatches(g)
    torch.manual_seed(20250718)
    model = PointNet2(depth=5).cuda().train()
    torch.manual_seed(31)
    avg_loss, ld = model.forward_hierarchical_streaming(dict(labels, mini_batches=iter(mbs)), return_loss=True, scaler=_FakeScaler())
    assert abs(avg_loss - float(g["avg_loss_f64"])) <= 2e-4 * abs(float(g["avg_loss_f64"]))
    assert abs(float(ld["offset_loss"].detach()) - float(g["offset_loss_f64"])) <= 2e-4 * abs(float(g["offset_loss_f64"]))
    assert abs(float(ld["semantic_loss"].detach()) - float(g["semantic_loss_f64"])) <= 2e-4 * abs(float(g["semantic_loss_f64"]))
    _check_grads(model, g, f"streaming overlap ({mode})")
    torch.manual_seed(32)
    with torch.no_grad():
        pred = model.forward_hierarchical_streaming(dict(labels, mini_batches=iter(mbs)), return_loss=False)
    close_to_reference(pred["offset_predictions"], g["pred_offsets"], g["pred_offsets_f64"], "overlap offsets", tol=2e-4)
    close_to_reference(pred["semantic_prediction_logits"], g["pred_logits"], g["pred_logits_f64"], "overlap logits", tol=2e-4)
    # and the opt-in averaging really is something else on this stream
    monkeypatch.setenv("PN2_OVERLAP", "average")
    torch.manual_seed(32)
    with torch.no_grad():
        avg = model.forward_hierarchical_streaming(dict(labels, mini_batches=iter(mbs)), return_loss=False)
    d = (avg["offset_predictions"] - pred["offset_predictions"]).abs().max().item()
    assert d > 1e-3 * np.abs(g["pred_offsets_f64"]).max(), d


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fused", "sequential"])
def test_hierarchical_overlapping_rasters_golden(mode, monkeypatch):
    """forward_hierarchical (autograd history through the accumulators) on the same overlapping stream: the loss of the
    averaged predictions and the gradients the reference's `t[ids] += v` expression implies (RefScatter / RefPut)."""
    helpers.load_pkg()
    monkeypatch.setenv("PN2_STREAMING", mode)
    monkeypatch.delenv("PN2_OVERLAP", raising=False)
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    g = gold("hierarchical_overlap_d5.npz")
    mbs, labels, _, n = _overlap_minibatches(gold("streaming_overlap_d5.npz"))
    torch.manual_seed(20250718)
    model = PointNet2(depth=5).cuda().train()
    torch.manual_seed(41)
    loss, ld = model.forward_hierarchical(dict(labels, mini_batches=iter(mbs)), return_loss=True)
    loss.backward()
    for key, val in (("loss", loss), ("offset_loss", ld["offset_loss"]), ("semantic_loss", ld["semantic_loss"])):
        assert abs(float(val.detach()) - float(g[key + "_f64"])) <= 2e-4 * abs(float(g[key + "_f64"])), key
    _check_grads(model, g, f"hierarchical overlap ({mode})")
