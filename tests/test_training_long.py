"""Training TRACE parity: 120 steps of the reference's own train_step (training.py:110-135 - compute_loss, clipping, AdamW /
Adam, nn.Embedding(max_norm=1) codes) on the reference's modules (tests/golden/make_golden_training_long.py ->
training_long.npz: per-step loss terms, parameter / code norms every 20 steps, the batches) replayed through this repo's
modules: the composite tier on the CPU (first steps, tight), the HIP training tier on the GPU (all 120 steps)."""
import math
import os

import numpy as np
import pytest
import torch

import _util as U
from nphm_amd.loss_functions import compute_loss

LAMBDAS = {"lat_reg": 0.01, "surf_sdf": 2.0, "normals": 0.3, "space_sdf": 0.01, "grad": 0.1, "anchors": 7.5,
           "symm_dist": 0.01, "middle_dist": 0.0}


def _replay(dev, n_steps, backend):
    fx = U.golden("training_long")
    net = U.build_identity(device=dev).train()
    assert U.state_hash(net) == str(fx["state_hash_init"])
    if backend == "composite":
        net.backend = "composite"
    net.train_backend = backend
    S = fx["codes_init"].shape[0]
    codes = torch.nn.Embedding(S, 1344, max_norm=1.0).to(dev)
    with torch.no_grad():
        codes.weight.copy_(torch.from_numpy(fx["codes_init"]))
    opt = torch.optim.AdamW(params=list(net.parameters()), lr=5e-4, weight_decay=0.01)
    opt_lat = torch.optim.Adam(list(codes.parameters()), lr=1e-3)
    keys = [str(k) for k in fx["keys"]]
    names = [str(n) for n in fx["snap_names"]]
    every = int(fx["snap_every"])
    trace, snaps = [], []

    def snapshot():
        sd = dict(net.named_parameters())
        return np.array([float(sd[n].detach().norm()) for n in names[:-1]] + [float(codes.weight.detach().norm())])
    for it in range(n_steps):
        if it % every == 0:
            snaps.append(snapshot())
        batch = {k[len("batches_"):]: torch.from_numpy(fx[k][it]) for k in fx if k.startswith("batches_")}
        batch["gt_anchors"] = torch.from_numpy(fx["pool_anchors"])
        batch["idx"] = torch.arange(S)[:, None]
        opt.zero_grad(); opt_lat.zero_grad()
        losses = compute_loss(batch, net, codes, dev)
        total = sum(LAMBDAS[k] * losses[k] for k in losses)
        total.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.1)
        torch.nn.utils.clip_grad_norm_(codes.parameters(), max_norm=0.1)
        opt.step(); opt_lat.step()
        trace.append([float(losses[k]) for k in keys] + [float(total)])
    snaps.append(snapshot())
    return fx, np.array(trace), np.stack(snaps), keys


def test_composite_tier_reproduces_the_reference_trace_cpu():
    fx, trace, snaps, keys = _replay(torch.device("cpu"), 3, "composite")
    ref = fx["trace"][:3]
    assert np.max(np.abs(trace - ref) / (np.abs(ref) + 1e-6)) < 2e-4
    assert np.max(np.abs(snaps[0] - fx["snaps"][0]) / fx["snaps"][0]) < 1e-6


@pytest.mark.gpu
def test_hip_training_tier_follows_the_reference_trace_120_steps():
    """The first 20 steps track the reference's CPU trace to 1e-4 of the loss (measured 8e-6), steps 20-60 to 1e-2
    (2e-3).  Afterwards the loop is chaotic (fresh batches every step, Adam normalising small gradients): the SAME loop
    on the composite tier of this GPU - the reference's arithmetic with other summation orders than its CPU run -
    stays within 1e-6 for 60 steps and is 5 % off by step 120; the HIP tier, whose gradients carry 5e-6 (fp32 operand
    storage) to 5e-5 (binary16 storage, what this 4 x 1000-point batch takes) of a tensor's largest entry from the pruned
    members and the split-bf16 sweeps - every sum in a fixed order since round 4, no float atomics - leaves the trace earlier
    and ends as far out (5-9 %).
    Asserted: those three bands, the mean loss of the last 20 steps within 5 %, every parameter norm within 3 %."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    fx, trace, snaps, keys = _replay(dev, fx_steps(), "hip")
    _, trace_c, snaps_c, _ = _replay(dev, fx_steps(), "composite")
    ref = fx["trace"]
    dev_h = np.abs(trace - ref)[:, -1] / ref[:, -1]
    dev_c = np.abs(trace_c - ref)[:, -1] / ref[:, -1]
    dev_hc = np.abs(trace - trace_c)[:, -1] / ref[:, -1]
    print("training trace over %d steps, total loss %.4f -> %.4f (reference on the CPU %.4f -> %.4f, composite tier on this GPU -> %.4f)"
          % (len(ref), trace[0, -1], trace[-1, -1], ref[0, -1], ref[-1, -1], trace_c[-1, -1]))
    print("relative deviation of the total loss from the reference's CPU trace: HIP tier first 20 steps %.2e, steps 20-60 %.2e, all %.2e | "
          "composite tier on this GPU %.2e / %.2e / %.2e | HIP vs composite %.2e" % (
              dev_h[:20].max(), dev_h[20:60].max(), dev_h.max(), dev_c[:20].max(), dev_c[20:60].max(), dev_c.max(), dev_hc.max()))
    assert dev_h[:20].max() < 1e-4 and dev_h[20:60].max() < 1e-2
    assert dev_h.max() < max(3.0 * dev_c.max(), 0.15)
    # mean loss over the last 20 steps (the noise of single steps averaged out)
    tail = lambda t: float(t[-20:, -1].mean())
    print("mean total loss of the last 20 steps: HIP %.5f, composite (GPU) %.5f, reference (CPU) %.5f" % (tail(trace), tail(trace_c), tail(ref)))
    assert abs(tail(trace) - tail(ref)) / tail(ref) < 5e-2
    snap_rel = np.abs(snaps - fx["snaps"]) / (fx["snaps"] + 1e-12)
    snap_rel_c = np.abs(snaps_c - fx["snaps"]) / (fx["snaps"] + 1e-12)
    print("parameter / code norms every 20 steps: max relative deviation HIP %.2e (composite on this GPU %.2e)" % (snap_rel.max(), snap_rel_c.max()))
    assert snap_rel.max() < 3e-2


def fx_steps():
    return int(U.golden("training_long")["trace"].shape[0])
