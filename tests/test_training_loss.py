"""The training loss of the identity decoder (nphm_amd.loss_functions, mirror of
src/NPHM/models/loss_functions.py:7-110) against tests/golden/training.npz = the reference's own
actual_compute_loss + loss.backward() on a seeded batch (make_golden_training.py).

CPU: the composite tier (same arithmetic as the reference) - loss terms to 1e-6, gradients to 1e-5 relative.
GPU: the HIP training tier (ident_train_kernel.hip) - loss terms to 1e-5, gradients to 5e-4 relative of the tensor's
largest entry (all 40 members; observed ~1e-5)."""
import numpy as np
import pytest
import torch

import _util as U
from NPHM.models.loss_functions import actual_compute_loss           # the path train.py imports

LAMBDAS = {"lat_reg": 0.01, "surf_sdf": 2.0, "normals": 0.3, "space_sdf": 0.01, "grad": 0.1, "anchors": 7.5,
           "symm_dist": 0.01, "middle_dist": 0.0}


def _step(net, g, dev):
    batch = {k[6:]: torch.from_numpy(g[k]).to(dev) for k in g if k.startswith("batch_")}
    lat = torch.from_numpy(g["lat"]).to(dev).requires_grad_()
    net.zero_grad(set_to_none=True)
    losses = actual_compute_loss(batch, net, lat)
    total = sum(LAMBDAS[k] * losses[k] for k in losses)
    total.backward()
    return losses, total, lat


def _check(net, g, losses, total, lat, tol_loss, tol_grad):
    assert U.state_hash(net) == str(g["state_hash"])
    for k, v in losses.items():
        assert abs(float(v) - float(g["loss_" + k])) <= tol_loss * max(1.0, abs(float(g["loss_" + k]))), k
    assert abs(float(total) - float(g["total"])) <= tol_loss
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    assert rel(lat.grad.cpu().numpy(), g["grad_lat"]) < tol_grad
    grads = dict(net.named_parameters())
    sets = g["sets"]
    for k in g:
        if k.startswith("grad_") and k[5:] in grads:
            mine = grads[k[5:]].grad.cpu().numpy()
            ref = g[k]
            if mine.shape != ref.shape:
                mine = mine[sets]
            assert rel(mine, ref) < tol_grad, k
    names = list(g["grad_names"])
    norms = np.array([float(grads[n].grad.norm()) for n in names])
    assert np.abs(norms - g["grad_norms"]).max() <= tol_grad * g["grad_norms"].max()


def test_loss_and_gradients_match_reference_cpu():
    g = U.golden("training")
    net = U.build_identity().train()
    net.backend = "composite"
    losses, total, lat = _step(net, g, torch.device("cpu"))
    assert set(losses) == set(LAMBDAS)
    _check(net, g, losses, total, lat, 1e-6, 1e-5)


@pytest.mark.gpu
def test_loss_and_gradients_match_reference_hip():
    dev = torch.device("cuda:0")
    g = U.golden("training")
    net = U.build_identity(device=dev).train()
    net.prune_tol = -1.0
    used = {}
    orig = net._forward_hip_train
    net._forward_hip_train = lambda *a, **k: used.setdefault("hip", True) and orig(*a, **k)
    losses, total, lat = _step(net, g, dev)
    assert used.get("hip"), "the HIP training tier did not run"
    _check(net, g, losses, total, lat, 1e-5, 5e-4)
    # default pruning (members below 1e-7 normalised blend weight dropped)
    net.prune_tol = 1e-7
    losses, total, lat = _step(net, g, dev)
    _check(net, g, losses, total, lat, 1e-5, 2e-3)
