"""CPU: the drop-in layout of INTEGRATION.md (option A) -- the package copied to Modules/pn2_amd and the reference's
three Modules/PointNet2 files replaced by re-export shims -- imports, builds every depth and keeps the reference's
state-dict keys and shapes."""
import importlib
import os
import shutil
import sys

import pytest

import helpers


@pytest.fixture()
def fake_reference(tmp_path):
    mod = tmp_path / "Modules"
    (mod / "PointNet2").mkdir(parents=True)
    (mod / "__init__.py").write_text("")
    (mod / "PointNet2" / "__init__.py").write_text("")
    shutil.copytree(helpers.PKG_DIR, mod / "pn2_amd", ignore=shutil.ignore_patterns("build", "build_diag", "__pycache__"))
    for m in ("pointnet2_utils", "blocks", "PointNet2"):
        (mod / "PointNet2" / f"{m}.py").write_text(f"from Modules.pn2_amd.PointNet2.{m} import *  # noqa: F401,F403\n")
    with open(mod / "PointNet2" / "PointNet2.py", "a") as f:
        f.write("from Modules.pn2_amd.PointNet2.PointNet2 import PointNet2\n")
    sys.path.insert(0, str(tmp_path))
    yield tmp_path
    sys.path.remove(str(tmp_path))
    for k in [k for k in sys.modules if k == "Modules" or k.startswith("Modules.")]:
        del sys.modules[k]


def test_reference_import_lines_work_through_the_shims(fake_reference):
    PointNet2 = importlib.import_module("Modules.PointNet2.PointNet2").PointNet2          # train_PointNet2.py:4
    blocks = importlib.import_module("Modules.PointNet2.blocks")
    utils = importlib.import_module("Modules.PointNet2.pointnet2_utils")
    for name in ("square_distance", "index_points", "farthest_point_sample", "query_ball_point", "sample_and_group",
                 "sample_and_group_all"):
        assert callable(getattr(utils, name)) and callable(getattr(blocks, name))       # blocks re-exports the L0 ops
    for name in ("ConvHead", "MLP", "PointNetSetAbstraction", "PointNetSetAbstractionMsg", "PointNetFeaturePropagation"):
        assert hasattr(blocks, name)
    n_params = {}
    for depth in (2, 3, 4, 5, 6):
        m = PointNet2(depth=depth)
        n_params[depth] = sum(p.numel() for p in m.parameters())
    # parameter counts measured on the imported reference
    assert n_params == {2: 164005, 3: 454181, 4: 983845, 5: 983845, 6: 1024853}
    sd = PointNet2(depth=5).state_dict()
    assert tuple(sd["sa1.mlp_convs.0.weight"].shape) == (32, 7, 1, 1)
    assert tuple(sd["fp1.mlp_convs.2.weight"].shape) == (128, 128, 1)
    assert tuple(sd["offset_linear.net.3.weight"].shape) == (3, 128, 1)
    for key in ("sa4.mlp_bns.2.running_var", "fp4.mlp_bns.0.num_batches_tracked", "semantic_linear.net.1.weight",
                "offset_linear.net.0.bias"):
        assert key in sd
    with pytest.raises(ValueError):
        PointNet2(depth=7)
