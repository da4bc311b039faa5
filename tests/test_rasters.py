"""Raster data path: rasters.py / csrc/raster.hip / DataLoading.RasterizedTreeSet against the numpy
restatement of the reference's loops (oracle/raster_port.py), element for element.  PARITY UNPINNED: the reference modules
import `fastprogress` (absent) and have no fixtures of their own."""
import numpy as np
import pytest
import torch

import helpers
from oracle import raster_port as R

pytestmark = pytest.mark.gpu


def _cloud(n, seed, grid=None):
    helpers.load_pkg()
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz, off, _ = gaussian_branch_tree(n, seed=seed)
    xyz = xyz.astype(np.float64)
    if grid:
        # Exact seams: move the cloud's minimum corner to the origin, so that the box bounds k * stride are the same numbers
        # in float64 and float32, and snap a share of the points onto them -- membership on a seam is then the same question
        # for the reference's float64 existence test and its float32 mask (lower bound inclusive, upper exclusive).
        xyz -= xyz.min(axis=0)
        xyz = xyz.astype(np.float32).astype(np.float64)
        xyz[0] = 0.0
        k = n // 5
        xyz[1:k] = np.round(xyz[1:k] / grid) * grid
    return xyz, off


@pytest.mark.parametrize("n,size,stride,mbs,seed,grid", [(6000, 1.0, 1.0, 10, 0, None), (6000, 1.0, 1.0, 7, 1, 1.0),
                                                         (3000, 1.0, 0.5, 6, 2, 0.5), (2500, 2.0, 2.0, 4, 3, None),
                                                         (2000, 0.75, 0.75, 60, 4, 0.75)])
def test_stream_equals_reference_loops(n, size, stride, mbs, seed, grid):
    from pn2_amd import rasters
    xyz64, off = _cloud(n, seed, grid)
    p32 = xyz64.astype(np.float32)
    feats = np.sin(0.3 * np.arange(n * 4)).astype(np.float32).reshape(n, 4)
    omask = (np.arange(n) % 3) != 0
    bounds = R.rasterize_clouds(xyz64, size, stride)
    ref_r = R.getitem_rasters(p32, feats, omask, bounds)
    keep = [r for r in ref_r if len(r["points"])]        # documented difference: a box must be non-empty in float32
    ref = R.collate_streaming(keep, mbs)
    dev = torch.device("cuda")
    stream = rasters.build_stream(torch.from_numpy(p32).to(dev), torch.from_numpy(feats).to(dev), torch.from_numpy(omask).to(dev),
                                  size, stride, mbs)
    # boxes that the float32 test fills although the float64 test left them empty cannot be compared with the loops; they
    # would show up as extra rasters
    assert stream.flat["rasters"] == len(keep), "box sets differ (float32 vs float64 seam?)"
    assert len(stream) == len(ref)
    for got, want in zip(stream, ref):
        for k in ("coords", "feats", "masks_pad", "masks_off", "point_ids"):
            g = got[k].cpu().numpy()
            assert g.shape == want[k].shape, k
            np.testing.assert_array_equal(g, want[k], err_msg=k)
    # every point lies in as many rasters as the float32 box test says, and the maximum corner may be in none
    count = np.bincount(np.concatenate([m["point_ids"] for m in ref]), minlength=n)
    if size == stride:
        assert count.max() <= 2 and (count == 1).mean() > 0.95
    # metadata: same boxes as rasterize_clouds wrote (for the boxes both keep)
    meta = rasters.raster_bounds_metadata(stream.flat["boxes"], stream.flat["dims"], stream.flat["bounds"], size)
    ref_lo = {tuple(np.round(b[0], 9)) for b, r in zip(bounds, ref_r) if len(r["points"])}
    assert {tuple(np.round(m["bounds"]["min"], 9)) for m in meta} == ref_lo


def test_minibatch_size_adjustment():
    from pn2_amd import rasters
    for n_r in range(2, 80):                               # n_r == 1: the reference's loop does not terminate
        for mb in (1, 2, 3, 10, 20, 60):
            orig = m = mb                                  # the reference's loop, verbatim (RasterizedTreeSet.py:395-404)
            while n_r % m == 1 and m > 1:
                m -= 1
                if m == 1:
                    m = orig
                    while n_r % m == 1:
                        m += 1
                    break
            assert rasters.adjusted_minibatch_size(n_r, mb) == m


def test_dataset_batch_runs_the_model_and_can_be_walked_twice(tmp_path):
    """RasterizedTreeSet_Hierarchical -> collate_fn_streaming -> forward_hierarchical_streaming: same predictions as the
    mini-batches built by the reference's loops; and the batch dict serves two models in a row."""
    from pn2_amd.DataLoading.RasterizedTreeSet import RasterizedTreeSet_Hierarchical, get_dataloader, rasterize_clouds
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    n = 12000
    xyz64, off = _cloud(n, 7)
    data = np.concatenate([xyz64, off.astype(np.float64), np.zeros((n, 1)), np.ones((n, 4))], axis=1)      # the 11-column label file
    path = tmp_path / "3_17_labeled.npy"
    np.save(path, data)
    js = tmp_path / "buffer.json"
    n_r = rasterize_clouds([str(path)], str(js), 1.0, 1.0, store_metadata=True)
    ds = RasterizedTreeSet_Hierarchical(str(js), training=False, minibatch_size=10, noise_distance=0.1)
    assert len(ds) == 1
    loader = get_dataloader(ds, 1, 0, False, ds.collate_fn_streaming)
    batch = next(iter(loader))
    assert batch["mini_batches"].flat["rasters"] == n_r
    torch.manual_seed(0)
    model = PointNet2(depth=5).cuda().eval()
    def run(b):
        torch.manual_seed(5)                                                            # same FPS start draws every time
        with torch.no_grad():
            return model.forward_hierarchical_streaming(b, return_loss=False)
    out1 = run(batch)
    out2 = run(batch)                                                                   # second model / second walk
    assert torch.equal(out1["offset_predictions"], out2["offset_predictions"]) and float(out1["offset_predictions"].abs().max()) > 0
    # the same tree through the reference-loop mini-batches
    p32 = xyz64.astype(np.float32)
    feats = np.ones((n, 4), np.float32)
    omask = np.linalg.norm(off, axis=1) <= 0.1
    ref = R.collate_streaming([r for r in R.getitem_rasters(p32, feats, omask, R.rasterize_clouds(xyz64, 1.0, 1.0)) if len(r["points"])], 10)
    mbs = [{k: torch.from_numpy(v).cuda() for k, v in m.items()} for m in ref]
    gen = (m for m in mbs)                                                              # a one-shot generator, like the reference's collate
    b2 = {"mini_batches": gen, "cloud_length": n}
    out3 = run(b2)
    out4 = run(b2)                                                                      # Q8: still sees the mini-batches
    assert torch.equal(out1["offset_predictions"], out3["offset_predictions"])
    assert torch.equal(out3["semantic_prediction_logits"], out4["semantic_prediction_logits"])


def test_dataset_honours_the_boxes_stored_in_the_json(tmp_path):
    """A JSON written with one raster_size / stride and a dataset constructed with other defaults: the reference reads the
    boxes from the JSON (RasterizedTreeSet.py:228-238), so the stored grid decides (round-2 advisor finding: the mirror used
    to re-grid with the constructor's 1.0 / 1.0)."""
    from pn2_amd.DataLoading.RasterizedTreeSet import RasterizedTreeSet_Hierarchical, rasterize_clouds
    n = 9000
    xyz64, off = _cloud(n, 11)
    data = np.concatenate([xyz64, off.astype(np.float64), np.zeros((n, 1)), np.ones((n, 4))], axis=1)
    path = tmp_path / "1_2_labeled.npy"
    np.save(path, data)
    js = tmp_path / "meta.json"
    n_r = rasterize_clouds([str(path)], str(js), 2.0, 1.0, store_metadata=True)            # overlapping 2 m boxes, stride 1 m
    ds = RasterizedTreeSet_Hierarchical(str(js), training=True, minibatch_size=10)            # constructor defaults: 1.0 / 1.0
    batch = ds.collate_fn_streaming([ds[0]])
    stream = batch["mini_batches"]
    assert stream.flat["rasters"] == n_r and not stream.disjoint
    want = R.collate_streaming([r for r in R.getitem_rasters(xyz64.astype(np.float32), np.ones((n, 4), np.float32),
                                                             np.linalg.norm(off, axis=1) <= 0.05, R.rasterize_clouds(xyz64, 2.0, 1.0))
                                if len(r["points"])], 10)
    assert len(stream) == len(want)
    for a, b in zip(stream, want):
        assert np.array_equal(a["point_ids"].cpu().numpy(), b["point_ids"])
        assert np.array_equal(a["coords"].cpu().numpy(), b["coords"])
