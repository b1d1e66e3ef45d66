"""CPU tests (no GPU): the oracle and the host modules against the golden fixtures produced by the
reference itself (tests/golden/make_golden.py).

Tolerances: the oracle and the reference are both fp32 with different GEMM accumulation orders
(numpy/OpenBLAS vs PyTorch/MKL, member-by-member vs bmm); observed differences are < 5e-7, the
bound asserted here is 2e-6 absolute.
"""
import numpy as np
import pytest
import torch

import _util as U
from oracle import nphm_oracle as O

TOL = 2e-6


# ---------------------------------------------------------------------------------------------
# seeded construction == reference weights
# ---------------------------------------------------------------------------------------------
def test_identity_weights_match_reference_hash():
    g = U.golden("nphm_identity")
    assert U.state_hash(U.build_identity()) == str(g["state_sha256"])
    g128 = U.golden("nphm_identity_pos128")
    assert U.state_hash(U.build_identity(pos_mlp_dim=128)) == str(g128["state_sha256"])


def test_deformation_and_npm_weights_match_reference_hash():
    assert U.state_hash(U.build_deformation()) == str(U.golden("deformation")["state_sha256"])
    assert U.state_hash(U.build_npm()) == str(U.golden("npm")["state_sha256"])


def test_state_dict_layout():
    sd = U.build_identity().state_dict()
    expect = {
        "ensembled_deep_sdf.lin0.weight": (24, 200, 99), "ensembled_deep_sdf.lin0.bias": (24, 200),
        "ensembled_deep_sdf.lin1.weight": (24, 101, 200), "ensembled_deep_sdf.lin1.bias": (24, 101),
        "ensembled_deep_sdf.lin2.weight": (24, 200, 200), "ensembled_deep_sdf.lin2.bias": (24, 200),
        "ensembled_deep_sdf.lin3.weight": (24, 200, 200), "ensembled_deep_sdf.lin3.bias": (24, 200),
        "ensembled_deep_sdf.lin4.weight": (24, 1, 200), "ensembled_deep_sdf.lin4.bias": (24, 1),
        "mlp_pos.0.weight": (256, 64), "mlp_pos.0.bias": (256,), "mlp_pos.2.weight": (256, 256),
        "mlp_pos.2.bias": (256,), "mlp_pos.4.weight": (117, 256), "mlp_pos.4.bias": (117,),
    }
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect      # 'anchors' must NOT be a key
    d = U.build_deformation().state_dict()
    assert tuple(d["compressor.0.weight"].shape) == (32, 1461)
    assert tuple(d["defDeepSDF.lin2.weight"].shape) == (277, 512)
    assert tuple(d["defDeepSDF.lin6.weight"].shape) == (3, 512)
    n = U.build_npm().state_dict()
    assert tuple(n["lin3.weight"].shape) == (509, 1024) and tuple(n["lin8.weight"].shape) == (1, 1024)


# ---------------------------------------------------------------------------------------------
# oracle vs reference goldens
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def identity_np():
    net = U.build_identity()
    return U.np_state(net), U.anchors_mean()


def test_oracle_identity_eval_and_train(identity_np):
    params, amean = identity_np
    g = U.golden("nphm_identity")
    xyz = g["xyz"][None]
    lat = np.repeat(g["lat"][None, None], xyz.shape[1], axis=1)
    sdf, anc = O.nphm_identity_forward(params, amean, xyz, lat, training=False)
    assert U.maxdiff(anc, g["anchors"]) < TOL
    assert U.maxdiff(sdf, g["sdf_eval"]) < TOL
    sdf_t, _ = O.nphm_identity_forward(params, amean, xyz, g["lat"][None, None], training=True)
    assert U.maxdiff(sdf_t, g["sdf_train"]) < TOL
    # the two modes differ only at the last point (reference quirk)
    d = np.abs(g["sdf_eval"] - g["sdf_train"]).reshape(-1)
    assert d[:-1].max() < TOL and d[-1] > 1e-3


def test_oracle_identity_batch2_and_per_point(identity_np):
    params, amean = identity_np
    g = U.golden("nphm_identity")
    sdf, anc = O.nphm_identity_forward(params, amean, g["b2_xyz"], g["b2_lat"][:, None], training=False)
    assert U.maxdiff(sdf, g["b2_sdf_eval"]) < TOL and U.maxdiff(anc, g["b2_anchors"]) < TOL
    sdf, _ = O.nphm_identity_forward(params, amean, g["pp_xyz"], g["pp_lat"], training=True)
    assert U.maxdiff(sdf, g["pp_sdf_train"]) < TOL


def test_oracle_get_logits_chunk_overwrite(identity_np):
    params, amean = identity_np
    g = U.golden("nphm_identity")
    res, chunk = int(g["grid_res"]), int(g["grid_chunk"])
    grid = O.create_grid_points_from_bounds(U.MINI, U.MAXI, res).astype(np.float32)[None]
    for training, key in ((False, "grid_logits_eval"), (True, "grid_logits_train")):
        fwd = lambda p, l, t=training: O.nphm_identity_forward(params, amean, p, l, training=t)
        vol = O.get_logits(fwd, g["lat"], grid, nbatch_points=chunk)
        assert U.maxdiff(vol, g[key]) < TOL
    hacked = O.hack_indices(res ** 3, chunk)
    diff = np.abs(g["grid_logits_eval"] - g["grid_logits_train"])
    mask = np.zeros(res ** 3, bool)
    mask[hacked] = True
    assert diff[~mask].max() < TOL
    assert list(hacked) == [499, 999, 1499, 1999, 2499, 2743]


def test_oracle_identity_stress_weights(identity_np):
    params, amean = identity_np
    g = U.golden("nphm_identity")
    p2 = dict(params)
    for i in range(5):
        p2[f"ensembled_deep_sdf.lin{i}.weight"] = params[f"ensembled_deep_sdf.lin{i}.weight"] * np.float32(g["stress_scale"])
    sdf, _ = O.nphm_identity_forward(p2, amean, g["xyz"][None, :1024], g["lat"][None, None], training=False)
    # sharper network: values up to ~1e2, compare relatively
    ref = g["stress_sdf_eval"]
    assert np.max(np.abs(sdf - ref) / (1.0 + np.abs(ref))) < 1e-5


def test_oracle_deformation_and_two_stage(identity_np):
    params, amean = identity_np
    g = U.golden("deformation")
    dparams = U.np_state(U.build_deformation())
    off, rest = O.deformation_forward(dparams, g["xyz"], g["lat"], g["anchors"])
    assert U.maxdiff(off, g["offsets"]) < TOL and U.maxdiff(rest, g["rest"]) < TOL
    res, chunk = int(g["grid_res"]), int(g["grid_chunk"])
    grid = O.create_grid_points_from_bounds(U.MINI, U.MAXI, res).astype(np.float32)[None]
    lat_id = g["lat"].reshape(-1)[:1344]
    f_shape = lambda p, l: O.nphm_identity_forward(params, amean, p, l, training=False)
    f_expr = lambda p, l: O.deformation_forward(dparams, p, l, g["anchors"])
    vol = O.get_logits_backward(f_shape, f_expr, lat_id, g["lat"].reshape(-1), grid, nbatch_points=chunk)
    assert U.maxdiff(vol, g["two_stage_logits"]) < TOL


def test_oracle_npm():
    g = U.golden("npm")
    params = U.np_state(U.build_npm())
    lat = np.repeat(g["lat"][None, None], g["xyz"].shape[1], axis=1)
    sdf = O.deepsdf_forward(params, "", g["xyz"], lat, nlayers=8)
    assert U.maxdiff(sdf, g["sdf"]) < 5e-6          # K=1024 GEMMs: slightly larger round-off
    res = int(g["grid_res"])
    grid = O.create_grid_points_from_bounds(U.MINI, U.MAXI, res).astype(np.float32)[None]
    fwd = lambda p, l: (O.deepsdf_forward(params, "", p, l, nlayers=8), None)
    assert U.maxdiff(O.get_logits(fwd, g["lat"], grid, nbatch_points=200), g["grid_logits"]) < 5e-6


def test_oracle_softplus_threshold():
    x = np.array([0.19, 0.2, 0.21, -0.5, 0.0], np.float32)
    y = O.softplus100(x)
    ref = torch.nn.functional.softplus(torch.from_numpy(x), beta=100).numpy()
    assert np.array_equal(y[:3], x[:3]) or U.maxdiff(y, ref) < 1e-9
    assert U.maxdiff(y, ref) < 1e-9


# ---------------------------------------------------------------------------------------------
# host modules (composite formulation, CPU, explicit opt-in) vs reference goldens
# ---------------------------------------------------------------------------------------------
def test_cpu_forward_refuses_without_opt_in():
    net = U.build_identity().eval()
    g = U.golden("nphm_identity")
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            net(torch.from_numpy(g["xyz"][None, :8]), torch.from_numpy(g["lat"][None, None]), None)


def test_composite_identity_matches_reference():
    net = U.build_identity()
    net.backend = "composite"
    g = U.golden("nphm_identity")
    xyz = torch.from_numpy(g["xyz"][None])
    lat = torch.from_numpy(g["lat"][None, None])
    with torch.no_grad():
        net.eval()
        sdf, anc = net(xyz, lat.repeat(1, xyz.shape[1], 1), None)
        assert U.maxdiff(sdf.numpy(), g["sdf_eval"]) < TOL and U.maxdiff(anc.numpy(), g["anchors"]) < TOL
        net.train()
        sdf, _ = net(xyz, lat, None)
        assert U.maxdiff(sdf.numpy(), g["sdf_train"]) < TOL
        sdf, _ = net(torch.from_numpy(g["pp_xyz"]), torch.from_numpy(g["pp_lat"]), None)
        assert U.maxdiff(sdf.numpy(), g["pp_sdf_train"]) < TOL
        net.eval()
        sdf, _ = net(torch.from_numpy(g["b2_xyz"]), torch.from_numpy(g["b2_lat"][:, None]), None)
        assert U.maxdiff(sdf.numpy(), g["b2_sdf_eval"]) < TOL
        # 2-D xyz is accepted (reference: unsqueeze)
        sdf2, _ = net(torch.from_numpy(g["xyz"][:16]), lat, None)
        assert sdf2.shape == (1, 16, 1)


def test_composite_autograd_first_and_second_order():
    net = U.build_identity()
    net.backend = "composite"
    net.train()
    g = U.golden("nphm_identity")
    xyz = torch.from_numpy(g["xyz"][None, :32]).clone().requires_grad_(True)
    lat = torch.from_numpy(g["lat"][None, None]).clone().requires_grad_(True)
    sdf, _ = net(xyz, lat, None)
    (gx,) = torch.autograd.grad(sdf.sum(), xyz, create_graph=True)
    assert gx.shape == xyz.shape and torch.isfinite(gx).all()
    loss = (gx.norm(dim=-1) - 1).pow(2).mean() + sdf.abs().mean()
    loss.backward()                                  # double backward (training path)
    assert lat.grad is not None and torch.isfinite(lat.grad).all()
    assert net.ensembled_deep_sdf.lin2.weight.grad is not None


def test_composite_deformation_and_npm():
    g = U.golden("deformation")
    dnet = U.build_deformation().eval()
    with pytest.raises(RuntimeError):                # CPU tensors: no silent non-HIP path
        with torch.no_grad():
            dnet(torch.from_numpy(g["xyz"]), torch.from_numpy(g["lat"]), torch.from_numpy(g["anchors"]))
    dnet.backend = "composite"
    with torch.no_grad():
        off, rest = dnet(torch.from_numpy(g["xyz"]), torch.from_numpy(g["lat"]), torch.from_numpy(g["anchors"]))
        assert U.maxdiff(off.numpy(), g["offsets"]) < TOL and U.maxdiff(rest.numpy(), g["rest"]) < TOL
        lat_rep = torch.from_numpy(g["lat"]).repeat(1, g["xyz"].shape[1], 1)
        anc_rep = torch.from_numpy(g["anchors"]).unsqueeze(1).repeat(1, g["xyz"].shape[1], 1, 1)
        off2, _ = dnet(torch.from_numpy(g["xyz"]), lat_rep, anc_rep)
        assert U.maxdiff(off2.numpy(), g["offsets"]) < TOL
    gn = U.golden("npm")
    npm = U.build_npm()
    npm.backend = "composite"
    with torch.no_grad():
        sdf, none = npm(torch.from_numpy(gn["xyz"]), torch.from_numpy(gn["lat"][None, None]))
        assert none is None and U.maxdiff(sdf.numpy(), gn["sdf"]) < 5e-6


def test_reference_attribute_surface():
    net = U.build_identity()
    for attr in ("lat_dim", "lat_dim_glob", "lat_dim_loc", "num_symm_pairs", "num_kps", "anchors", "mlp_pos",
                 "input_dim", "out_dim", "ensembled_deep_sdf"):
        assert hasattr(net, attr)
    assert net.lat_dim == 1344 and net.num_kps == 39
    d = U.build_deformation()
    assert d.lat_dim_expr == 200 and d.out_dim == 4 and d.lat_dim == 232 and hasattr(d, "compressor")
    with pytest.raises(ValueError):
        import nphm_amd
        nphm_amd.DeformationNetwork("nope", 200, 32, 64, 32, 39, None, 512)


def test_member_point_lists_cover_exactly_the_kept_pairs():
    """Host logic of the autograd tier: the (row, member) tile table handed to the member-centric
    kernels lists every (point, member) pair the pruning rule keeps exactly once, in tiles of <= 64
    points of one (row, member), and the dropped weights respect the budget 40 * prune_tol."""
    from nphm_amd.ensembled_deepsdf import _member_point_lists
    rng = np.random.default_rng(3)
    B, N, A = 2, 300, 40
    anchors = torch.from_numpy(np.repeat(U.anchors_mean()[None], B, 0) + 0.01 * rng.standard_normal((B, 39, 3))).float()
    xyz = torch.from_numpy(rng.uniform(U.MINI, U.MAXI, size=(B, N, 3))).float()
    for tol in (1e-7, 1e-4, -1.0):
        what, tiles, plist = _member_point_lists(anchors, xyz, tol, A)
        kept = what > 0
        seen = torch.zeros(B, N, A, dtype=torch.int32)
        for b, k, off, cnt in tiles.tolist():
            assert 0 < cnt <= 64
            pts = plist[off:off + cnt].long()
            seen[b, pts, k] += 1
        if tol < 0:
            assert bool((seen == 1).all())
        else:
            assert torch.equal(seen > 0, kept) and int(seen.max()) == 1
            # what the rule dropped sums to <= 40 * tol per point (full weights recomputed here)
            d = (anchors[:, None] - xyz[:, :, None]).norm(dim=3) + 1e-5
            w = torch.exp(-(d * d) / 0.01)
            w_bg = float(np.exp(-20.0))
            full = torch.cat([w, torch.full_like(w[..., :1], w_bg)], 2) / (w.sum(2, keepdim=True) + w_bg + 1e-6)
            dropped = (full * (~kept)).sum(2)
            assert float(dropped.max()) <= 40 * tol * (1 + 1e-5)
