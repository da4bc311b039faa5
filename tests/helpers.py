"""Shared test helpers (used by tests/ and by tests/golden/make_golden.py)."""
import importlib.util
import math
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
PKG_DIR = os.path.join(REPO, "extracting-tree-morphology-from-point-clouds_amd")

if REPO not in sys.path:
    sys.path.insert(0, REPO)


def load_pkg():
    """Import the (hyphenated) product package under the alias ``pn2_amd``."""
    name = "pn2_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def closed_form_init(model):
    """Deterministic, RNG-free parameter pattern written BY NAME ORDER, so the reference model (when the
    fixture is generated) and the product model (when it is tested) hold identical weights without
    shipping a state dict.  weights: A/sqrt(fan_in) * sin(0.37 i + k); biases/BN beta: 0.1 sin;
    BN gamma: 1 + 0.2 sin."""
    import torch
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(model.named_parameters(), key=lambda kv: kv[0])):
            n = p.numel()
            i = torch.arange(n, dtype=torch.float64)
            s = torch.sin(0.37 * i + float(k))
            if p.dim() >= 2:
                fan_in = p.shape[1]
                v = 1.5 / math.sqrt(fan_in) * s
            elif "bns" in name or "bn_blocks" in name or name.endswith("net.1.weight") or name.endswith("net.1.bias"):
                v = (1.0 + 0.2 * s) if name.endswith("weight") else 0.1 * s
            else:
                v = 0.1 * s
            p.copy_(v.to(torch.float32).view_as(p))


def raster_batch(n_real, n_pad_to, seed=0, cube=None):
    """A raster-like batch: real points of the synthetic tree inside dense 1 m cubes, zero padded.
    Returns coords [B,3,N] f32, masks_pad [B,N] bool, offsets [B,N,3]."""
    load_pkg()
    from pn2_amd.synthetic import gaussian_branch_tree, rasterize
    xyz, off, _ = gaussian_branch_tree(262144, seed=seed)
    rasters = sorted(rasterize(xyz), key=len, reverse=True)
    B = len(n_real)
    coords = np.zeros((B, 3, n_pad_to), np.float32)
    offs = np.zeros((B, n_pad_to, 3), np.float32)
    mask = np.zeros((B, n_pad_to), bool)
    for b, n in enumerate(n_real):
        r = rasters[b if cube is None else cube[b]][:n]
        assert len(r) == n, (len(r), n)
        coords[b, :, :n] = xyz[r].T
        offs[b, :n] = off[r]
        mask[b, :n] = True
    return coords, mask, offs


def is_pre_bn_bias(name):
    """Bias of a conv that feeds a train-mode BatchNorm: its gradient is zero in exact arithmetic (the batch
    mean removes any constant), so what the reference stores there is rounding noise and cannot be compared
    relatively."""
    import re
    return bool(re.search(r"(mlp_convs\.\d+|conv_blocks\.\d+\.\d+|net\.0)\.bias$", name))
