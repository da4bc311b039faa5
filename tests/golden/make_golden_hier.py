"""Golden vector for PointNet2.forward_hierarchical (the NON-streaming raster mode, reference PointNet2.py:329-394),
produced by running the REFERENCE's own method on the CPU:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hier.py
Like make_golden.make_streaming: the method hard-codes device="cuda", which is mapped to "cpu" in torch.zeros /
Tensor.to for the duration of the call.  Same 6-raster tree, weights and mini-batches as streaming_d5.npz; stored: the
loss of the averaged predictions and the parameter-gradient norms of loss.backward(), in fp32 and with float64 layer
arithmetic."""
import contextlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402
import torch  # noqa: E402


def main(size=2.0, stride=2.0, name="hierarchical_d5.npz", first=0):
    MG.helpers.load_pkg()
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz, off, _ = gaussian_branch_tree(20000, seed=5)
    rasters = MG.overlap_rasters(xyz, size, stride, first=first)
    feats_all = MG.sinpat((len(xyz), 4), 7)
    n = int(len(xyz))
    sem_lab = (np.arange(n) % 3 == 0).astype(np.int64)

    def mini_batches():
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
            moff = (np.arange(len(ids)) % 5) != 2
            yield {k_: torch.from_numpy(v) for k_, v in (("coords", coords), ("feats", fts), ("masks_pad", mpad),
                                                         ("masks_off", moff), ("point_ids", ids))}

    zeros, to = torch.zeros, torch.Tensor.to

    def zeros_cpu(*a, **k):
        if k.get("device") == "cuda":
            k["device"] = "cpu"
        return zeros(*a, **k)

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        return to(self, *a, **k)

    out = {}
    torch.zeros, torch.Tensor.to = zeros_cpu, to_cpu
    try:
        for f64 in (False, True):
            tag = "_f64" if f64 else ""
            with (MG._f64_layers() if f64 else contextlib.nullcontext()):
                torch.manual_seed(MG.WEIGHT_SEED)
                model = MG.RP.PointNet2(depth=5)
                model.train()
                batch = {"cloud_length": n, "mini_batches": mini_batches(), "semantic_labels": torch.from_numpy(sem_lab)[:, None],
                         "offset_labels": torch.from_numpy(off)}
                torch.manual_seed(41)
                loss, ld = model.forward_hierarchical(batch, return_loss=True)
                loss.backward()
                names, l2, _ = MG.grad_summary(model)
                out.update({"loss" + tag: np.float64(loss.item()), "offset_loss" + tag: np.float64(ld["offset_loss"].item()),
                            "semantic_loss" + tag: np.float64(ld["semantic_loss"].item()), "grad_names": names, "grad_l2" + tag: l2})
    finally:
        torch.zeros, torch.Tensor.to = zeros, to
    assert MG._fp_ties[0] == 0
    MG.save(name, **out)
    print({k: (float(v) if np.ndim(v) == 0 else v.shape) for k, v in out.items()})


if __name__ == "__main__":
    if "--overlap" in sys.argv:      # the training default: size 2.0, stride 1.0 (train_PointNet2.py:84-85,109)
        main(2.0, 1.0, "hierarchical_overlap_d5.npz", first=MG.OVERLAP_FIRST)
    else:
        main()
