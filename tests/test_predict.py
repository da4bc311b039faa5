"""BASELINE configs[4] at test size: predict_all_trees + kNN-to-QSM on a small synthetic forest, sharded by tree.
The two-shard run returns exactly what the single-process run returns for the same trees (no collective, no coupling)."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _forest(n_trees, n_points):
    helpers.load_pkg()
    from pn2_amd.synthetic import gaussian_branch_tree
    trees, qsms = [], []
    for t in range(n_trees):
        xyz, off, seg = gaussian_branch_tree(n_points, seed=40 + t)
        trees.append(xyz.astype(np.float64))
        rng = np.random.default_rng(t)
        m = 30
        start = xyz[rng.integers(0, n_points, m)] + off[:m] * 0
        end = start + rng.normal(size=(m, 3)) * 0.5
        qsms.append({"startX": start[:, 0], "startY": start[:, 1], "startZ": start[:, 2], "endX": end[:, 0], "endY": end[:, 1],
                     "endZ": end[:, 2], "radius": rng.uniform(0.02, 0.2, m), "ID": np.arange(m) + 100 * t})
    return trees, qsms


def test_sharded_forest_prediction_equals_single_process():
    helpers.load_pkg()
    from pn2_amd import predict
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    trees, qsms = _forest(5, 8000)
    torch.manual_seed(0)
    m_off = PointNet2(depth=5).cuda().eval()
    m_noise = PointNet2(depth=5).cuda().eval()

    def run(rank, world):                                     # per-tree FPS start stream: independent of the sharding
        return predict.predict_forest(m_off, m_noise, trees, qsms, rank=rank, world=world, seed=1000, minibatch_size=10)

    whole = run(0, 1)
    shards = {}
    for r in range(2):
        shards.update(run(r, 2))
    assert sorted(shards) == sorted(whole) == list(range(5))
    for i in whole:
        for k in whole[i]:
            np.testing.assert_array_equal(whole[i][k], shards[i][k], err_msg=f"tree {i} {k}")
        r = whole[i]
        assert r["pred_full"].shape == (8000, 7) and set(np.unique(r["pred_full"][:, 6])) <= {0.0, 1.0}
        assert len(r["executed_cloud"]) == int((r["pred_full"][:, 6] == 0).sum())
        assert len(r["qsm_ids"]) == len(r["executed_cloud"]) and (r["qsm_ids"] // 100 == i).all()
        assert np.isfinite(r["qsm_offsets"]).all() and (r["qsm_distance"] >= 0).all()


def test_xyz_only_cloud_predicts_the_same_through_predict_tree_and_the_dataset():
    """An xyz-only cloud is padded with eight zero columns by the reference dataset (RasterizedTreeSet.py:207-211), i.e. its
    network features are ZEROS; predict_tree must feed the same features as the dataset + forward_hierarchical_streaming path
    (round-2 advisor finding: it fed ones)."""
    helpers.load_pkg()
    from pn2_amd import predict
    from pn2_amd.DataLoading.RasterizedTreeSet import RasterizedTreeSet_Hierarchical
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz, _, _ = gaussian_branch_tree(6000, seed=77)
    cloud = xyz.astype(np.float64)
    torch.manual_seed(0)
    model = PointNet2(depth=5).cuda().eval()
    torch.manual_seed(5)
    res = predict.predict_tree(model, model, cloud, minibatch_size=10)
    ds = RasterizedTreeSet_Hierarchical([], training=False, minibatch_size=10)
    item = ds.from_array(cloud)
    assert item["features"].shape == (6000, 4) and not item["features"].any()
    batch = ds.collate_fn_streaming([item])
    torch.manual_seed(5)
    with torch.no_grad():
        noise = model.forward_hierarchical_streaming(batch, return_loss=False)["semantic_prediction_logits"]
        out = model.forward_hierarchical_streaming(batch, return_loss=False)["offset_predictions"]
    np.testing.assert_array_equal(res["pred_full"][:, 3:6], out.cpu().numpy().astype(np.float64))
    np.testing.assert_array_equal(res["pred_full"][:, 6], torch.argmax(noise, 1).cpu().numpy().astype(np.float64))
    with pytest.raises(ValueError):
        predict.predict_tree(model, model, np.zeros((10, 5)))
