"""DeformationNetwork in its other conditioning modes ('glob_only', 'expr_only', 'interpolate', 'GNN'; the reference's
deepSDF.py:118-239) and DeepSDF with positional encoding (deepSDF.py:14-37, 64-73) against outputs of the reference's own
modules (tests/golden/def_modes.npz, tests/golden/make_golden_def_modes.py): the mirrored modules reproduce the reference's
seeded weights bit for bit (SHA-256 of the state_dict) and its outputs - on the CPU through the composite tier, on the GPU
through whatever tier serves the call: the fused kernel for conditioning that is constant along the points (one row per batch
entry: 'glob_only', 'expr_only', 'GNN' called the way the lattice / fitting drivers call them), the composite tier with its
hidden products on the dense kernels for per-point conditioning ('interpolate', positional encoding)."""
import numpy as np
import pytest
import torch

import _util as U
import nphm_amd

MODES = ("glob_only", "expr_only", "interpolate", "GNN")
TOL = 1e-5          # (the north star's bar is 1e-4)


def _build(mode, seed, device):
    anchors = torch.from_numpy(U.anchors_mean()).float().unsqueeze(0).unsqueeze(0)
    torch.manual_seed(seed)
    return nphm_amd.DeformationNetwork(mode=mode, lat_dim_expr=200, lat_dim_id=32, lat_dim_glob_shape=64, lat_dim_loc_shape=32,
                                       n_loc=39, anchors=anchors, hidden_dim=512, nlayers=6, input_dim=3, out_dim=3).to(device).eval()


def _posenc(device):
    torch.manual_seed(31)
    return nphm_amd.DeepSDF(lat_dim=40, hidden_dim=256, nlayers=6, geometric_init=True, out_dim=1, input_dim=3,
                            num_freq_bands=4).to(device).eval()


@pytest.mark.parametrize("mode", MODES)
def test_modes_reproduce_the_reference_on_the_cpu(mode):
    g = U.golden("def_modes")
    net = _build(mode, 10 + MODES.index(mode), "cpu")
    net.backend = "composite"                     # (the CPU is an explicit opt-in: the product path fails loudly without a ROCm device)
    assert U.state_hash(net) == str(g[f"{mode}_sha256"])
    xyz, lat, anc = torch.from_numpy(g["xyz"]), torch.from_numpy(g[f"{mode}_lat"]), torch.from_numpy(g["anchors"])
    n = xyz.shape[1]
    with torch.no_grad():
        off, rest = net(xyz, lat.repeat(1, n, 1), anc.repeat(1, n, 1, 1))
        off1, rest1 = net(xyz, lat, anc)                                  # one conditioning row for all points: same field
    assert U.maxdiff(off.numpy(), g[f"{mode}_offsets"]) < 1e-6 and U.maxdiff(rest.numpy(), g[f"{mode}_rest"]) < 1e-6
    assert U.maxdiff(off1.numpy(), g[f"{mode}_offsets"]) < 1e-6 and U.maxdiff(rest1.numpy(), g[f"{mode}_rest"]) < 1e-6


def test_positional_encoding_reproduces_the_reference_on_the_cpu():
    g = U.golden("def_modes")
    net = _posenc("cpu")
    net.backend = "composite"
    assert U.state_hash(net) == str(g["posenc_sha256"])
    xyz, cond = torch.from_numpy(g["xyz"]), torch.from_numpy(g["posenc_cond"])
    with torch.no_grad():
        y, _ = net(xyz, cond.repeat(1, xyz.shape[1], 1))
    assert U.maxdiff(y.numpy(), g["posenc_out"]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_modes_against_the_reference_on_the_gpu(mode):
    g = U.golden("def_modes")
    dev = torch.device("cuda:0")
    net = _build(mode, 10 + MODES.index(mode), dev)
    assert U.state_hash(net) == str(g[f"{mode}_sha256"])
    xyz, lat, anc = (torch.from_numpy(g[k]).to(dev) for k in ("xyz", f"{mode}_lat", "anchors"))
    n = xyz.shape[1]
    calls = {"hip": 0}
    orig = net.defDeepSDF.forward_hip

    def spy(*a, **k):
        calls["hip"] += 1
        return orig(*a, **k)

    net.defDeepSDF.forward_hip = spy
    with torch.no_grad():
        off1, rest1 = net(xyz, lat, anc)                                  # the drivers' form: one conditioning row
        served = calls["hip"]
        off, rest = net(xyz, lat.repeat(1, n, 1), anc.repeat(1, n, 1, 1))   # the reference's per-point form
    # conditioning constant along the points -> the fused kernel; 'interpolate' conditions every point on its position
    assert served == (0 if mode == "interpolate" else 1), (mode, served)
    for o, r in ((off1, rest1), (off, rest)):
        assert U.maxdiff(o.cpu().numpy(), g[f"{mode}_offsets"]) < TOL and U.maxdiff(r.cpu().numpy(), g[f"{mode}_rest"]) < TOL


@pytest.mark.gpu
def test_positional_encoding_against_the_reference_on_the_gpu():
    g = U.golden("def_modes")
    dev = torch.device("cuda:0")
    net = _posenc(dev)
    xyz, cond = torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["posenc_cond"]).to(dev)
    with torch.no_grad():
        y, _ = net(xyz, cond.repeat(1, xyz.shape[1], 1))
        y1, _ = net(xyz, cond)
    assert U.maxdiff(y.cpu().numpy(), g["posenc_out"]) < TOL and U.maxdiff(y1.cpu().numpy(), g["posenc_out"]) < TOL
