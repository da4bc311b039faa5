"""Element-wise BatchNorm-parameter gradients of the depth-3 model fixture, from the IMPORTED reference:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_d3_bn.py
Round 1 compared only gradient NORMS there, and sa2.mlp_bns.2.bias sat at 3x the reference's own fp32 noise
(VERDICT r1, weak #3).  This fixture stores, for the inputs / seeds of model_d3.npz, d(loss*50)/d(gamma, beta) of every
BatchNorm of sa2 and sa3 element by element, from the reference in fp32 and with float64 layer arithmetic, so that the
HIP path can be judged per channel (tests/test_round2.py::test_depth3_bn_gradients_elementwise)."""
import contextlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402  (imports the reference, installs the CPU work-arounds)
import torch  # noqa: E402


def main():
    g = np.load(os.path.join(HERE, "model_d3.npz"))
    batch = {k: torch.from_numpy(g[k]) for k in ["coords", "feats", "masks_pad", "masks_off", "semantic_labels", "offset_labels"]}
    out = {}
    for f64 in (False, True):
        torch.manual_seed(int(g["weight_seed"]))
        model = MG.RP.PointNet2(depth=3).train()
        with (MG._f64_layers() if f64 else contextlib.nullcontext()):
            torch.manual_seed(int(g["torch_seed"]))
            loss, _ = model(batch, return_loss=True)
            (loss * 50).backward()
        assert abs(float(loss) - float(g["loss_f64" if f64 else "loss"])) <= 1e-6 * abs(float(g["loss"]))
        for n, p in model.named_parameters():
            if n.startswith(("sa2.mlp_bns", "sa3.mlp_bns")):
                out[("g64__" if f64 else "g__") + n] = p.grad.detach().numpy().copy()
    MG.save("model_d3_bn_grads.npz", **out)


if __name__ == "__main__":
    main()
