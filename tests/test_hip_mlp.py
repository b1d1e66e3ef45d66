"""GPU parity tests of the fused dense skip-MLP kernel (libnphm_amd.so: nphm_mlp_*), i.e. the
reference's DeepSDF / DeformationNetwork (src/NPHM/models/deepSDF.py) and the two-stage drivers
get_logits_backward / deform_mesh (src/NPHM/models/reconstruction.py:28-88), against
  * the golden fixtures produced by the reference's own PyTorch modules (tests/golden/make_golden.py),
  * the numpy oracle (oracle/nphm_oracle.py) on seeded inputs.
Tolerance: the north star's 1e-4 absolute (TOL_BAR); the split-bf16 path is expected ~1e-6."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import _util as U
import nphm_amd
from nphm_amd import reconstruction as R
from oracle import nphm_oracle as O

pytestmark = pytest.mark.gpu
TOL_BAR = 1e-4
TOL_TIGHT = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _exact(net):
    """the value+Jacobian / Broyden / saving launches on the three-term product (the exactness tests below compare them with
    autograd to 1e-6 .. 2e-5; the default, fit_numerics = "auto", runs calibrated two-term layers there: its own test)"""
    (net.defDeepSDF if hasattr(net, "defDeepSDF") else net).fit_numerics = "f16x3"
    return net


@pytest.fixture(scope="module")
def dnet(dev):
    return _exact(U.build_deformation(device=dev).eval())


@pytest.fixture(scope="module")
def npm(dev):
    return _exact(U.build_npm(device=dev).eval())


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _spy(module, name):
    called = {}
    orig = getattr(module, name)

    def wrapper(*a, **k):
        called["n"] = called.get("n", 0) + 1
        return orig(*a, **k)
    setattr(module, name, wrapper)
    return called, lambda: setattr(module, name, orig)


def test_hip_tier_is_what_runs(dnet, npm, dev):
    g = U.golden("deformation")
    assert dnet.defDeepSDF.hip_supported() and npm.hip_supported()
    called, restore = _spy(dnet.defDeepSDF, "forward_hip")
    try:
        with torch.no_grad():
            dnet(_t(g["xyz"], dev), _t(g["lat"], dev), _t(g["anchors"], dev))
    finally:
        restore()
    assert called.get("n") == 1
    # training mode adds per-point conditioning noise (deepSDF.py:220-221) -> composite tier
    dnet.train()
    called, restore = _spy(dnet.defDeepSDF, "forward_hip")
    try:
        with torch.no_grad():
            dnet(_t(g["xyz"], dev), _t(g["lat"], dev), _t(g["anchors"], dev))
    finally:
        restore()
        dnet.eval()
    assert not called


def test_deformation_golden(dnet, dev):
    g = U.golden("deformation")
    xyz, lat, anc = _t(g["xyz"], dev), _t(g["lat"], dev), _t(g["anchors"], dev)
    with torch.no_grad():
        off, rest = dnet(xyz, lat, anc)
        assert off.shape == (1, xyz.shape[1], 3) and rest.shape == (1, xyz.shape[1], 1)
        e = U.maxdiff(off.cpu().numpy(), g["offsets"])
        print(f"deformation max abs err vs reference: {e:.3e}")
        assert e < TOL_TIGHT and U.maxdiff(rest.cpu().numpy(), g["rest"]) < TOL_TIGHT
        # get_logits-style repeated latent / per-point anchors (row 0 is what counts)
        n = xyz.shape[1]
        off2, _ = dnet(xyz, lat.repeat(1, n, 1), anc.unsqueeze(1).repeat(1, n, 1, 1))
        assert torch.equal(off, off2)
        # canonical points = x + offsets, fused
        can = dnet.canonical_points(xyz, lat, anc)
        assert U.maxdiff(can.cpu().numpy(), g["xyz"] + g["offsets"]) < TOL_TIGHT


def test_npm_golden(npm, dev):
    g = U.golden("npm")
    with torch.no_grad():
        sdf, none = npm(_t(g["xyz"], dev), _t(g["lat"][None, None], dev))
    e = U.maxdiff(sdf.cpu().numpy(), g["sdf"])
    print(f"NPM max abs err vs reference: {e:.3e}")
    assert none is None and sdf.shape == (1, g["xyz"].shape[1], 1) and e < 2e-5
    res = int(g["grid_res"])
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).to(dev, dtype=torch.float)[None]
    called, restore = _spy(R, "evaluate_grid_mlp")
    try:
        vol = R.get_logits(npm, _t(g["lat"], dev), grid, nbatch_points=200)
    finally:
        restore()
    assert called.get("n") == 1 and vol.dtype == np.float32
    assert U.maxdiff(vol, g["grid_logits"]) < 2e-5


@pytest.mark.parametrize("n", [1, 31, 64, 65, 1000])
def test_ragged_sizes_and_batch_rows_vs_oracle(dnet, dev, n):
    rng = np.random.default_rng(n)
    B = 3
    xyz = rng.uniform(-0.6, 0.6, size=(B, n, 3)).astype(np.float32)
    lat = (0.3 * rng.standard_normal((B, 1, 1544))).astype(np.float32)
    anc = (U.anchors_mean()[None] + 0.01 * rng.standard_normal((B, 39, 3))).astype(np.float32)
    ref, _ = O.deformation_forward(U.np_state(dnet), xyz, lat, anc)
    with torch.no_grad():
        off, _ = dnet(_t(xyz, dev), _t(lat, dev), _t(anc, dev))
    assert U.maxdiff(off.cpu().numpy(), ref) < TOL_TIGHT
    # 2-D xyz is accepted (reference: no batch axis handling in DeepSDF, DeformationNetwork unsqueezes)
    with torch.no_grad():
        off1, _ = dnet(_t(xyz[0], dev), _t(lat[:1], dev), _t(anc[:1], dev))
    assert U.maxdiff(off1.cpu().numpy(), ref[:1]) < TOL_TIGHT


def test_grid_kernel_equals_points_kernel_bitwise(dnet, npm, dev):
    axes = R.grid_axes(U.MINI, U.MAXI, (5, 7, 11))
    pts = torch.from_numpy(np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(1, -1, 3)).to(dev)
    g = U.golden("deformation")
    mlp, cond = R._expr_condition(dnet, _t(g["lat"], dev), _t(g["anchors"], dev), dev)
    a = R.evaluate_grid_mlp(mlp, cond, axes)
    b = mlp.forward_hip(pts, cond)
    assert torch.equal(a, b[0])
    # x-slabs are slices of the full volume (multi-GPU sharding invariant)
    s0 = R.evaluate_grid_mlp(mlp, cond, axes, x_range=(0, 2), add_input=True)
    s1 = R.evaluate_grid_mlp(mlp, cond, axes, x_range=(2, 5), add_input=True)
    full = R.evaluate_grid_mlp(mlp, cond, axes, add_input=True)
    assert torch.equal(torch.cat([s0, s1]), full)
    assert U.maxdiff((full - a).cpu().numpy(), pts[0].cpu().numpy()) < 1e-6
    gn = U.golden("npm")
    v = R.evaluate_grid_mlp(npm, _t(gn["lat"][None], dev), axes)
    w = npm.forward_hip(pts, _t(gn["lat"][None], dev))
    assert torch.equal(v, w[0])


def test_two_stage_golden_and_oracle(dnet, dev):
    g = U.golden("deformation")
    inet = U.build_identity(device=dev).eval()
    inet.prune_tol = -1.0
    res, chunk = int(g["grid_res"]), int(g["grid_chunk"])
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).to(dev, dtype=torch.float)[None]
    lat_id = _t(g["lat"].reshape(-1)[:1344], dev)
    lat_ex = _t(g["lat"].reshape(-1), dev)
    anc = _t(g["anchors"], dev)
    called, restore = _spy(R, "evaluate_grid_two_stage")
    try:
        vol = R.get_logits_backward(inet, dnet, lat_id, lat_ex, grid, nbatch_points=chunk, anchors=anc)
    finally:
        restore()
    assert called.get("n") == 1
    e = U.maxdiff(vol, g["two_stage_logits"])
    print(f"two-stage (deformation -> identity) max abs err vs reference: {e:.3e}")
    assert e < TOL_BAR
    # non-lattice point set: points kernels, same chunk rule
    perm = torch.randperm(res ** 3, generator=torch.Generator().manual_seed(5)).to(dev)
    vol_p = R.get_logits_backward(inet, dnet, lat_id, lat_ex, grid[:, perm], nbatch_points=chunk, anchors=anc)
    hacked = O.hack_indices(res ** 3, chunk)
    keep = np.ones(res ** 3, bool); keep[hacked] = False
    inv = np.empty(res ** 3, np.int64); inv[perm.cpu().numpy()] = np.arange(res ** 3)
    keep_both = keep & keep[inv]
    assert U.maxdiff(vol_p[inv][keep_both], g["two_stage_logits"][keep_both]) < TOL_BAR
    # default anchors = the identity net's predicted anchors (config 3 of BASELINE.json)
    vol_d, can = R.evaluate_grid_two_stage(inet, dnet, lat_id, lat_ex, R.grid_axes(U.MINI, U.MAXI, 12),
                                           hack_chunk=0, return_canonical=True)
    params, amean = U.np_state(inet), U.anchors_mean()
    pts = O.create_grid_points_from_bounds(U.MINI, U.MAXI, 12).astype(np.float32)[None]
    _, anc_pred = O.nphm_identity_forward(params, amean, pts[:, :1], g["lat"][:, :, :1344], training=True)
    off, _ = O.deformation_forward(U.np_state(dnet), pts, g["lat"], anc_pred)
    assert U.maxdiff(can.cpu().numpy(), pts[0] + off[0]) < TOL_TIGHT
    ref, _ = O.nphm_identity_forward(params, amean, (pts + off).astype(np.float32), g["lat"][:, :, :1344], training=True)
    assert U.maxdiff(vol_d.cpu().numpy(), ref.reshape(-1)) < TOL_BAR


def test_deform_mesh_matches_oracle(dnet, dev):
    g = U.golden("deformation")
    rng = np.random.default_rng(2)
    verts = rng.uniform(-0.4, 0.4, size=(7001, 3))
    mesh = SimpleNamespace(vertices=verts, faces=np.zeros((1, 3), np.int64))
    lat_id = _t(g["lat"][..., :1344], dev)
    lat_ex = _t(g["lat"][..., 1344:], dev)
    out = R.deform_mesh(mesh, dnet, lat_ex, _t(g["anchors"], dev), lat_rep_shape=lat_id)
    off, _ = O.deformation_forward(U.np_state(dnet), verts.astype(np.float32)[None], g["lat"], g["anchors"])
    assert U.maxdiff(np.asarray(out.vertices), verts.astype(np.float32) + off[0]) < TOL_TIGHT


def test_weight_update_invalidates_pack_and_composite_agrees(dev):
    d = U.build_deformation(device=dev).eval()
    g = U.golden("deformation")
    xyz, lat, anc = _t(g["xyz"][:, :200], dev), _t(g["lat"], dev), _t(g["anchors"], dev)
    with torch.no_grad():
        a, _ = d(xyz, lat, anc)
        d.defDeepSDF.lin4.weight.mul_(1.5)
        d.defDeepSDF.lin0.bias.add_(0.01)
        b, _ = d(xyz, lat, anc)
        assert not torch.equal(a, b)
        d.backend = "composite"
        c, _ = d(xyz, lat, anc)
    assert U.maxdiff(b.cpu().numpy(), c.cpu().numpy()) < TOL_TIGHT


def test_stress_weights_relative(dev):
    """Sharper networks (weights x2): activations leave the softplus knee, outputs grow; compare
    relatively against the fp32 oracle."""
    d = U.build_deformation(device=dev).eval()
    n = U.build_npm(device=dev).eval()
    rng = np.random.default_rng(9)
    xyz = rng.uniform(-0.6, 0.6, size=(1, 777, 3)).astype(np.float32)
    g, gn = U.golden("deformation"), U.golden("npm")
    with torch.no_grad():
        for m in (d.defDeepSDF, n):
            for i in range(m.num_layers - 1):
                getattr(m, f"lin{i}").weight.mul_(2.0)
        off, _ = d(_t(xyz, dev), _t(g["lat"], dev), _t(g["anchors"], dev))
        sdf, _ = n(_t(xyz, dev), _t(gn["lat"][None, None], dev))
    ref, _ = O.deformation_forward(U.np_state(d), xyz, g["lat"], g["anchors"])
    rel = np.max(np.abs(off.cpu().numpy() - ref) / (1.0 + np.abs(ref)))
    lat = np.repeat(gn["lat"][None, None], xyz.shape[1], axis=1)
    refn = O.deepsdf_forward(U.np_state(n), "", xyz, lat, nlayers=8)
    reln = np.max(np.abs(sdf.cpu().numpy() - refn) / (1.0 + np.abs(refn)))
    print(f"stress x2: deformation max rel err {rel:.3e} (|ref| up to {np.abs(ref).max():.2f}), "
          f"NPM {reln:.3e} (|ref| up to {np.abs(refn).max():.2f})")
    assert rel < TOL_BAR and reln < TOL_BAR


def test_flattened_batch_with_segmented_conditioning(dnet, dev):
    """The reference's root finder flattens a batch of observations into ONE row of points whose
    conditioning is constant on equal segments (iterative_root_finding.py:137-139); the HIP tier
    re-views it as batch rows."""
    rng = np.random.default_rng(4)
    S, n = 5, 333
    xyz = rng.uniform(-0.5, 0.5, size=(S, n, 3)).astype(np.float32)
    lat = (0.3 * rng.standard_normal((S, 1, 1544))).astype(np.float32)
    lat[:, :, :1344] = lat[:1, :, :1344]                      # shared identity code, per-row expression
    anc = np.repeat(U.anchors_mean()[None], S, 0).astype(np.float32)
    called, restore = _spy(dnet.defDeepSDF, "forward_hip")
    try:
        with torch.no_grad():
            a, _ = dnet(_t(xyz, dev), _t(lat, dev), _t(anc, dev))
            flat_lat = _t(np.repeat(lat, n, 1).reshape(1, S * n, 1544), dev)
            flat_anc = _t(np.repeat(anc[:, None], n, 1).reshape(1, S * n, 39, 3), dev)
            b, _ = dnet(_t(xyz.reshape(1, S * n, 3), dev), flat_lat, flat_anc)
            # irregular segments are not re-viewed: composite tier, same values
            irregular = flat_lat.clone()
            irregular[0, 7] = irregular[0, -1]
            c, _ = dnet(_t(xyz.reshape(1, S * n, 3), dev), irregular, flat_anc)
            # neighbouring segments with the same code merge into one run: still segment-constant
            lat_m = lat.copy(); lat_m[1] = lat_m[0]
            a_m, _ = dnet(_t(xyz, dev), _t(lat_m, dev), _t(anc, dev))
            b_m, _ = dnet(_t(xyz.reshape(1, S * n, 3), dev), _t(np.repeat(lat_m, n, 1).reshape(1, S * n, 1544), dev),
                          flat_anc)
    finally:
        restore()
    assert called.get("n") == 4 and torch.equal(a_m.reshape(1, S * n, 3), b_m)
    assert torch.equal(a.reshape(1, S * n, 3), b)
    ref, _ = O.deformation_forward(U.np_state(dnet), xyz, lat, anc)
    assert U.maxdiff(b.cpu().numpy().reshape(S, n, 3), ref) < TOL_TIGHT
    keep = np.ones(S * n, bool); keep[7] = False
    assert U.maxdiff(c.cpu().numpy()[0, keep], ref.reshape(-1, 3)[keep]) < TOL_TIGHT


def test_fused_jacobian_matches_autograd(dnet, npm, dev):
    """nphm_mlp_eval_points_jvp (forward-mode tangents through the same GEMMs) vs the reference's
    definition of jac: one forward + three autograd VJPs (diff_operators.py:26-54) on the composite
    tier, and the flattened-batch form the root finder uses."""
    from nphm_amd import diff_operators as D
    rng = np.random.default_rng(12)
    S, n = 5, 213
    xyz = rng.uniform(-0.5, 0.5, size=(S, n, 3)).astype(np.float32)
    lat = (0.3 * rng.standard_normal((S, 1, 1544))).astype(np.float32)
    lat[:, :, :1344] = lat[:1, :, :1344]
    anc = np.repeat(U.anchors_mean()[None], S, 0).astype(np.float32)
    x = _t(xyz, dev)
    posed, J = dnet.jacobian(x, _t(lat, dev), _t(anc, dev))
    assert posed.shape == (S, n, 3) and J.shape == (S, n, 3, 3) and not J.requires_grad
    # the inverse from the same launch (ABI 10): the separate inverse3x3 launch on that Jacobian, up to the contraction of
    # the adjugate's products into FMAs
    posed_i, J_i, J_inv = dnet.jacobian(x, _t(lat, dev), _t(anc, dev), inverse=True)
    assert torch.equal(posed_i, posed) and torch.equal(J_i, J) and J_inv.shape == (S, n, 3, 3)
    assert float((J_inv - D.inverse3x3(J)).abs().max()) < 1e-6
    assert float((J_inv @ J - torch.eye(3, device=dev)).abs().max()) < 1e-5
    with torch.no_grad():
        off, _ = dnet(x, _t(lat, dev), _t(anc, dev))
    assert U.maxdiff((x + off).cpu().numpy(), posed.cpu().numpy()) < 1e-6
    dnet.backend = "composite"
    try:
        J_ref = D.jac(dnet, x.clone(), _t(lat, dev).repeat(1, n, 1), _t(anc, dev).unsqueeze(1).repeat(1, n, 1, 1))
    finally:
        dnet.backend = "hip"
    e = U.maxdiff(J.cpu().numpy(), J_ref.cpu().numpy())
    print(f"fused Jacobian vs autograd: max abs err {e:.3e} (|J - I| up to {float((J_ref - torch.eye(3, device=dev)).abs().max()):.3f})")
    assert e < 2e-5
    # through jac() itself, in the flattened-batch form (segment-constant conditioning)
    called, restore = _spy(dnet.defDeepSDF, "forward_hip_jvp")
    try:
        J_flat = D.jac(dnet, x.reshape(1, S * n, 3).clone(), _t(np.repeat(lat, n, 1).reshape(1, S * n, 1544), dev),
                       _t(np.repeat(anc[:, None], n, 1).reshape(1, S * n, 39, 3), dev))
    finally:
        restore()
    assert called.get("n") == 1 and torch.equal(J_flat.reshape(S, n, 3, 3), J)
    # NPM SDF: the spatial gradient (surface normal direction)
    gn = U.golden("npm")
    q = _t(gn["xyz"], dev)
    out = npm.forward_hip_jvp(q, _t(gn["lat"][None], dev))
    npm.backend = "composite"
    try:
        qg = q.clone().requires_grad_(True)
        sdf, _ = npm(qg, _t(gn["lat"][None, None], dev))
        (g_ref,) = torch.autograd.grad(sdf.sum(), qg)
    finally:
        npm.backend = "hip"
    assert U.maxdiff(out[:, :, 0, 0].cpu().numpy(), sdf[..., 0].detach().cpu().numpy()) < 2e-5
    assert U.maxdiff(out[:, :, 1:, 0].cpu().numpy(), g_ref.cpu().numpy()) < 5e-5


def test_fused_broyden_matches_the_python_solver(dev):
    """nphm_mlp_broyden vs iterative_root_finding.broyden (the reference's algorithm) driven by the
    same HIP forwards: same roots, same residuals, same convergence set — also for a field that is
    sharp enough (weights x3) for some points to need many iterations / diverge."""
    from nphm_amd import iterative_root_finding as IRF
    rng = np.random.default_rng(21)
    for scale, n_min_valid in ((1.0, 0.99), (1.6, 0.0)):
        d = U.build_deformation(device=dev).eval()
        with torch.no_grad():
            for i in range(d.defDeepSDF.num_layers - 1):
                getattr(d.defDeepSDF, f"lin{i}").weight.mul_(scale if i < d.defDeepSDF.num_layers - 2 else scale ** 2)
        S, n = 3, 500
        obs = _t(rng.uniform(-0.4, 0.4, size=(S, n, 3)).astype(np.float32), dev)
        lat = (0.3 * rng.standard_normal((S, 1, 1544))).astype(np.float32)
        lat[:, :, :1344] = lat[:1, :, :1344]
        cond = _t(np.repeat(lat, n, 1), dev)
        anc = _t(np.repeat(U.anchors_mean()[None, None], S, 0).repeat(n, 1).astype(np.float32), dev)
        torch.manual_seed(0)
        xc_f, res_f = IRF.search(obs, cond, d, anc, multi_corresp=False)          # fused
        fused = d.broyden
        d.broyden = lambda *a, **k: None                                           # python solver, HIP forwards
        try:
            torch.manual_seed(0)
            xc_p, res_p = IRF.search(obs, cond, d, anc, multi_corresp=False)
            # the fused solver evaluating its own start residual (search hands it x_init + F(x_init) of the Jacobian launch)
            d.broyden = lambda *a, **k: fused(*a, **{kk: v for kk, v in k.items() if kk != "posed_init"})
            torch.manual_seed(0)
            xc_n, res_n = IRF.search(obs, cond, d, anc, multi_corresp=False)
        finally:
            d.broyden = fused
        assert torch.equal(res_n["valid_ids"], res_f["valid_ids"]) and float((xc_n - xc_f).abs().max()) <= 1e-6
        vf, vp = res_f["valid_ids"], res_p["valid_ids"]
        frac = float(vp.float().mean())
        both = (vf & vp)
        if not bool(both.any()):
            both = torch.ones_like(both)          # nothing converged: compare the final iterates
        print(f"weights x{scale}: converged {frac:.3f} (python) / {float(vf.float().mean()):.3f} (fused), "
              f"mismatching flags {int((vf != vp).sum())}, max |dx| on common roots "
              f"{float((xc_f - xc_p)[both].abs().max()):.2e}")
        assert xc_f.shape == (S, n, 3) and vf.shape == (S, n) and frac >= n_min_valid
        assert int((vf != vp).sum()) <= max(2, int(0.01 * S * n))
        assert float((xc_f - xc_p)[both].abs().max()) < 1e-5
        assert float((res_f["diff"] - res_p["diff"]).reshape(S, n)[both].abs().max()) < 2e-6
        # a converged root satisfies x_c + F(x_c) = x_obs
        with torch.no_grad():
            off, _ = d(xc_f, cond, anc)
        assert float((xc_f + off - obs)[vf].norm(dim=-1).max()) < 5e-6


def test_two_stage_full_size_256_cubed_properties(dnet, dev):
    """BASELINE.json configs[2] at full size: deformation -> identity on 256^3, checked through
    determinism, slab invariance and a random subsample against the points kernels / the oracle."""
    res = 256
    inet = U.build_identity(device=dev).eval()
    g = U.golden("deformation")
    lat_id = _t(g["lat"].reshape(-1)[:1344], dev)
    lat_ex = _t(g["lat"].reshape(-1), dev)
    axes = R.grid_axes(U.MINI, U.MAXI, res)
    vol, can = R.evaluate_grid_two_stage(inet, dnet, lat_id, lat_ex, axes, hack_chunk=0, return_canonical=True)
    assert vol.shape == (res ** 3,) and can.shape == (res ** 3, 3) and bool(torch.isfinite(vol).all())
    slab = R.evaluate_grid_two_stage(inet, dnet, lat_id, lat_ex, axes, hack_chunk=0, x_range=(96, 104))
    assert torch.equal(slab, vol.view(res, -1)[96:104].reshape(-1))
    anchors = inet.prepare_latent(lat_id[None])[2]
    # 4 608 voxels stratified by the identity field's blend regime AT THE LATTICE POINT (near an anchor / mid-field /
    # far field; the canonical points are displaced by the deformation): all of them against the numpy oracle
    keep = U.stratified_voxels(axes, anchors[0].cpu().numpy(), 1536, seed=1)
    ax, ay, az = axes
    pts = np.stack([ax[keep // (res * res)], ay[(keep // res) % res], az[keep % res]], -1).astype(np.float32)
    with torch.no_grad():
        off, _ = dnet(_t(pts[None], dev), lat_ex[None, None], anchors)
    k = torch.from_numpy(keep).to(dev)
    # (the lattice launch runs the calibrated per-layer tiers - DeepSDF.numerics_target = 5e-6 on its sample, single-term
    # layers included - the 4 608-point call below the three-term product everywhere)
    assert U.maxdiff(can[k].cpu().numpy(), pts + off[0].cpu().numpy()) <= 2.0 * dnet.defDeepSDF.numerics_target
    rep = dnet.defDeepSDF.last_numerics
    print("two-stage 256^3: deformation layers", rep)
    assert rep is not None and rep["verified_err"] <= dnet.defDeepSDF.two_pass_target
    off_o, _ = O.deformation_forward(U.np_state(dnet), pts[None], g["lat"], anchors.cpu().numpy())
    ref, _ = O.nphm_identity_forward(U.np_state(inet), U.anchors_mean(), (pts[None] + off_o).astype(np.float32),
                                     g["lat"][:, :, :1344], training=True)
    err = np.abs(vol[k].cpu().numpy() - ref.reshape(-1))
    print("two-stage 256^3 vs oracle, max |err| near-anchor / mid-field / far-field: %.2e / %.2e / %.2e"
          % (err[:1536].max(), err[1536:3072].max(), err[3072:].max()))
    assert float(err.max()) < 1e-5


def test_npm_full_size_64_cubed_vs_reference_sequence(npm, dev):
    """BASELINE.json configs[0] at its stated size: NPM global DeepSDF on the 64^3 lattice through get_logits
    (chunk 25 000), against the reference's PyTorch operation sequence evaluated in full on the host
    (oracle/torch_reference.py: chunked get_logits, fp32), plus determinism and slab invariance."""
    from oracle import torch_reference as T
    res = 64
    g = U.golden("npm")
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None]
    lat = _t(g["lat"], dev)
    vol = R.get_logits(npm, lat, grid.to(dev), nbatch_points=25000)
    assert vol.shape == (res ** 3,) and vol.dtype == np.float32 and np.isfinite(vol).all()
    assert np.array_equal(vol, R.get_logits(npm, lat, grid.to(dev), nbatch_points=25000))          # deterministic
    axes = R.grid_axes(U.MINI, U.MAXI, res)
    slab = R.evaluate_grid_mlp(npm, lat[None], axes, x_range=(24, 32))
    assert np.array_equal(slab.reshape(-1).cpu().numpy(), vol.reshape(res, -1)[24:32].reshape(-1))
    sd = {k: v.detach().cpu() for k, v in npm.state_dict().items()}
    ref = T.get_logits(lambda p, l: T.deepsdf_forward(sd, "", p, l, nlayers=8), torch.from_numpy(g["lat"])[None, None],
                       grid, 25000)
    e = U.maxdiff(vol, ref)
    print(f"NPM 64^3 max abs err vs the reference op sequence: {e:.3e} (max |sdf| {np.abs(ref).max():.3f})")
    assert e < 2e-5


# ---------------------------------------------------------------------------------------------
# hand-written backward of the deformation backbone w.r.t. its conditioning (fitting loop)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 64, 777, 4133, 9000])
def test_backward_wrt_conditioning_matches_autograd(dev, n):
    """d/d(lat_rep, anchors) of sum(offsets * cotangent) through DeformationNetwork with frozen parameters and
    detached points: HIP forward + mlp_bwd_kernel vs autograd through the composite formulation.  (2 x n points: up to
    2 x 4096 run as 32-point workgroups - one round of 256 CUs -, beyond as 64-point ones; 4133 ends inside the first
    tile of a 64-point group, 777 inside the first tile of an odd 32-point workgroup's group.)"""
    net = _exact(U.build_deformation(device=dev).eval())
    for p in net.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(11)
    xyz = ((torch.rand(2, n, 3, generator=g) - 0.5) * 0.6).to(dev)
    cot = torch.randn(2, n, 3, generator=g).to(dev)
    lat0 = (torch.randn(2, 1, 1544, generator=g) * 0.05).to(dev)
    anc0 = (torch.from_numpy(U.anchors_mean()).float()[None] + 0.01 * torch.randn(2, 39, 3, generator=g)).to(dev)

    def run(backend):
        net.backend = backend
        lat = lat0.clone().requires_grad_(True)
        anc = anc0.clone().requires_grad_(True)
        off, rest = net(xyz, lat, anc)
        ((off * cot).sum() + 0.3 * rest.sum()).backward()
        return off.detach(), lat.grad, anc.grad

    called, restore = _spy(net.defDeepSDF, "forward_hip_cond_grad")
    try:
        off_h, gl_h, ga_h = run("hip")
    finally:
        restore()
    assert called.get("n", 0) == 1                     # the HIP autograd tier served the call
    off_c, gl_c, ga_c = run("composite")
    net.backend = "hip"
    assert U.maxdiff(off_h.cpu(), off_c.cpu()) < TOL_TIGHT
    for a, b in ((gl_h, gl_c), (ga_h, ga_c)):
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) < 2e-5 * scale + 1e-9


def test_backward_tier_selection(dev):
    """points or parameters that require grad, or per-point conditioning, keep the composite tier; double backward
    through the HIP tier raises"""
    net = U.build_deformation(device=dev).eval()
    g = U.golden("deformation")
    xyz, lat, anc = _t(g["xyz"], dev), _t(g["lat"], dev), _t(g["anchors"], dev)
    called, restore = _spy(net.defDeepSDF, "forward_hip_cond_grad")
    try:
        net(xyz, lat.clone().requires_grad_(True), anc)                       # parameters still require grad
        for p in net.parameters():
            p.requires_grad_(False)
        net(xyz.clone().requires_grad_(True), lat.clone().requires_grad_(True), anc)    # d/dxyz wanted
        net(xyz, lat.repeat(1, xyz.shape[1], 1).requires_grad_(True), anc)              # per-point conditioning
        assert called.get("n", 0) == 0
        l2 = lat.clone().requires_grad_(True)
        off, _ = net(xyz, l2, anc)
        assert called.get("n", 0) == 1
        (gl,) = torch.autograd.grad(off.sum(), l2, create_graph=True)
        with pytest.raises(RuntimeError):
            gl.sum().backward()
    finally:
        restore()


@pytest.mark.parametrize("n", [5, 64, 1000])
def test_posed_and_jacobian_in_one_launch(dev, n):
    """DeformationNetwork.posed_and_jacobian = forward (+ x) and jacobian at the same points from ONE launch that also
    leaves the backward's state: same values as the two separate calls, same conditioning gradient as autograd through
    the composite formulation."""
    net = _exact(U.build_deformation(device=dev).eval())
    for p in net.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(21)
    xyz = ((torch.rand(3, n, 3, generator=g) - 0.5) * 0.6).to(dev)
    cot = torch.randn(3, n, 3, generator=g).to(dev)
    lat0 = (torch.randn(3, 1, 1544, generator=g) * 0.05).to(dev)
    anc = (torch.from_numpy(U.anchors_mean()).float()[None] + 0.01 * torch.randn(3, 39, 3, generator=g)).to(dev)
    lat = lat0.clone().requires_grad_(True)
    posed, J = net.posed_and_jacobian(xyz, lat, anc)
    assert posed.shape == (3, n, 3) and J.shape == (3, n, 3, 3) and not J.requires_grad
    with torch.no_grad():
        off, _ = net(xyz, lat0, anc)
    posed_ref, J_ref = net.jacobian(xyz, lat0, anc)
    assert torch.equal(posed.detach(), posed_ref) and torch.equal(J, J_ref)
    assert U.maxdiff((xyz + off).cpu(), posed.detach().cpu()) < 1e-6
    (posed * cot).sum().backward()
    net.backend = "composite"
    lat_c = lat0.clone().requires_grad_(True)
    off_c, _ = net(xyz, lat_c, anc)
    ((off_c + xyz) * cot).sum().backward()
    net.backend = "hip"
    scale = float(lat_c.grad.abs().max())
    assert float((lat.grad - lat_c.grad).abs().max()) < 2e-5 * scale + 1e-9
    # the implicit root x_c = x - J^-1 (F - F.detach()) in the same launch, -J^-T inside the backward launch (ABI 10): the values
    # are the points, the conditioning gradient is that of the two-piece form (posed_and_jacobian + the root's own backward)
    from nphm_amd import fitting as F
    lat_a, lat_b = lat0.clone().requires_grad_(True), lat0.clone().requires_grad_(True)
    xc = net.implicit_root(xyz, lat_a, anc)
    assert xc is not None and torch.equal(xc.detach(), xyz) and xc.requires_grad
    (xc * cot).sum().backward()
    posed_b, _, jinv_b = net.posed_and_jacobian(xyz, lat_b, anc, inverse=True)
    (F._ImplicitRootFn.apply(xyz, posed_b, jinv_b) * cot).sum().backward()
    assert float((lat_a.grad - lat_b.grad).abs().max()) < 1e-6 * float(lat_b.grad.abs().max()) + 1e-12


# ---- round 4: operand format and two-term layers of the plain evaluation (include/nphm_amd.h NPHM_MLP_*) ------------------
@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_deformation_golden_both_formats(dnet, dev, precision):
    """Either operand format reproduces the reference fixture; the split-f16 one (22 product bits) is the default."""
    g = U.golden("deformation")
    xyz, lat, anc = _t(g["xyz"], dev), _t(g["lat"], dev), _t(g["anchors"], dev)
    mlp = dnet.defDeepSDF
    keep = (mlp.precision, mlp.numerics)
    try:
        mlp.precision, mlp.numerics = precision, "fixed"
        with torch.no_grad():
            off, _ = dnet(xyz, lat, anc)
    finally:
        mlp.precision, mlp.numerics = keep
    err = U.maxdiff(off.cpu().numpy(), g["offsets"])
    print(f"deformation golden, {precision}: max|hip - reference| = {err:.3e}")
    assert err < (5e-7 if precision == "f16x3" else 2e-6)


def _big_points(dev, n=1 << 18, seed=3):
    gen = torch.Generator().manual_seed(seed)
    lo, hi = torch.tensor(U.MINI), torch.tensor(U.MAXI)
    return (torch.rand(1, n, 3, generator=gen) * (hi - lo) + lo).to(dev)


def test_two_term_layers_fixed_mask_and_auto(dev):
    """Two-term products (weights rounded to f16) on every hidden layer: a small, measurable perturbation; `auto` picks a
    mask whose measured error stays inside two_pass_target and reports it; small evaluations never use the tier."""
    dnet = U.build_deformation(device=dev).eval()
    mlp = dnet.defDeepSDF
    cond = torch.randn(1, mlp.lat_dim, device=dev) * 0.05
    xyz = _big_points(dev)
    with torch.no_grad():
        mlp.numerics, mlp.two_pass_mask = "fixed", 0
        ref = mlp.forward_hip(xyz, cond)
        mlp.two_pass_mask = mlp._hidden_mask()
        two = mlp.forward_hip(xyz, cond)
        e_two = float((two - ref).abs().max())
        scale = float(ref.abs().max())
        print(f"all hidden layers two-term: max|d| = {e_two:.3e} on outputs up to {scale:.3e}")
        assert 0.0 < e_two < 1e-4
        mlp.numerics, mlp.two_pass_mask = "auto", 0
        auto = mlp.forward_hip(xyz, cond)
        rep = mlp.last_numerics
        e_auto = float((auto - ref).abs().max())
        print("auto:", rep, f"full-set error {e_auto:.3e}")
        assert rep is not None and rep["calibrated_here"] and rep["err"] <= mlp.two_pass_target
        assert e_auto <= 2.0 * mlp.two_pass_target
        # a second conditioning: the calibrated mask is verified on a sample of that call before it is used
        cond2 = torch.randn(1, mlp.lat_dim, device=dev) * 0.3
        mlp.numerics = "fixed"; mlp.two_pass_mask = 0
        ref2 = mlp.forward_hip(xyz, cond2)
        mlp.numerics = "auto"
        auto2 = mlp.forward_hip(xyz, cond2)
        rep2 = mlp.last_numerics
        assert rep2["calibrated_here"] is False or rep2.get("recalibrated_for_conditioning")
        assert float((auto2 - ref2).abs().max()) <= 2.0 * mlp.two_pass_target
        # small calls: three-term everywhere, bitwise the fixed mode
        small = xyz[:, :5000].contiguous()
        mlp.numerics = "fixed"; mlp.two_pass_mask = 0
        a = mlp.forward_hip(small, cond)
        mlp.numerics = "auto"
        b = mlp.forward_hip(small, cond)
        assert torch.equal(a, b)


def test_single_term_layers_fixed_and_auto(dev):
    """Single-term layers rn(x) wh (ABI 9): pinned on every hidden layer (the 128-points-per-workgroup variant), pinned on
    some (the generic kernel skips their lo products), and what `auto` settles on - each against the three-term product;
    the lattice launch is bitwise the points launch in every tier."""
    dnet = U.build_deformation(device=dev).eval()
    mlp = dnet.defDeepSDF
    cond = torch.randn(1, mlp.lat_dim, device=dev) * 0.05
    xyz = _big_points(dev)
    hid = mlp._hidden_mask()
    with torch.no_grad():
        mlp.numerics, mlp.two_pass_mask, mlp.single_mask = "fixed", 0, 0
        ref = mlp.forward_hip(xyz, cond)
        errs = {}
        tail = 1 << (mlp.nlayers - 1)                        # the last hidden layer
        for name, two, one in (("all", 0, hid), ("all but the last, two-term", tail, hid & ~tail), ("layers 1,3 single / rest two-term", hid, 0b1010),
                               ("layer 2 single / rest three-term", 0, 0b100)):
            mlp.two_pass_mask, mlp.single_mask = two, one
            out = mlp.forward_hip(xyz, cond)
            errs[name] = float((out - ref).abs().max())
            # ragged tail: N not a multiple of 128
            a = mlp.forward_hip(xyz[:, :1000 + 77].contiguous(), cond)
            assert torch.equal(a, out[:, :1077])
        print("single-term tiers against the three-term product:", errs, "outputs up to", float(ref.abs().max()))
        assert 0.0 < errs["layer 2 single / rest three-term"] < errs["all"] < 5e-5
        # (the 128-point variant with the last hidden layer in two point halves: the layer that lost most with its lo half)
        assert 0.0 < errs["all but the last, two-term"] < errs["all"]
        # lattice launch = points launch, bit for bit, in the 128-point variant too
        axes = R.grid_axes(U.MINI, U.MAXI, 20)
        pts = torch.from_numpy(np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)).to(dev)
        mlp.two_pass_mask, mlp.single_mask = 0, hid
        vol = R.evaluate_grid_mlp(mlp, cond, axes, add_input=False)
        assert torch.equal(vol.reshape(-1, 3), mlp.forward_hip(pts, cond).reshape(-1, 3))
        # auto: whatever it picks stays within twice the target on the full set, and reports it
        mlp.numerics, mlp.two_pass_mask, mlp.single_mask = "auto", 0, 0
        auto = mlp.forward_hip(xyz, cond)
        rep = mlp.last_numerics
        e_auto = float((auto - ref).abs().max())
        print("auto:", rep, f"full-set error {e_auto:.3e}")
        assert rep["err"] <= mlp.numerics_target and e_auto <= 2.0 * mlp.numerics_target
        assert rep["single_mask"] != 0                     # (the seeded net takes single-term layers at 5e-6)
        # switched off: the two-term tier of rounds 3-4
        mlp.allow_single_term = False
        auto2 = mlp.forward_hip(xyz, cond)
        assert mlp.last_numerics["single_mask"] == 0 and float((auto2 - ref).abs().max()) <= 2.0 * mlp.numerics_target


def test_two_term_tier_on_sharp_weights_stays_inside_target(dev):
    """Weights x 2.5 (the stress model of trained sharpness): whatever mask `auto` settles on, the evaluation stays within
    twice the target of the three-term product - the tier shrinks instead of the error growing."""
    dnet = U.build_deformation(device=dev).eval()
    mlp = dnet.defDeepSDF
    with torch.no_grad():
        for i in range(mlp.num_layers - 2):
            getattr(mlp, f"lin{i}").weight.mul_(2.5 if i else 1.5)
        mlp.invalidate_pack()
        cond = torch.randn(1, mlp.lat_dim, device=dev) * 0.2
        xyz = _big_points(dev, seed=5)
        mlp.numerics, mlp.two_pass_mask = "fixed", 0
        ref = mlp.forward_hip(xyz, cond)
        mlp.numerics = "auto"
        auto = mlp.forward_hip(xyz, cond)
        rep = mlp.last_numerics
        err = float((auto - ref).abs().max())
        print(f"weights x2.5: outputs up to {float(ref.abs().max()):.3e}, auto mask {rep['mask']:#x}, error {err:.3e}", rep)
        assert err <= 2.0 * mlp.two_pass_target


def test_npm_formats_agree_with_fixture(npm, dev):
    g = U.golden("npm")
    keep = (npm.precision, npm.numerics)
    try:
        for precision, tol in (("f16x3", 1e-6), ("bf16x3", 2e-6)):
            npm.precision, npm.numerics = precision, "fixed"
            with torch.no_grad():
                out, _ = npm(_t(g["xyz"], dev), _t(g["lat"][None, None], dev))
            err = U.maxdiff(out.cpu().numpy(), g["sdf"])
            print(f"npm golden, {precision}: {err:.3e}")
            assert err < tol
    finally:
        npm.precision, npm.numerics = keep


def test_fit_tier_two_term_layers_against_the_three_term_launches(dev):
    """fit_numerics = "auto" (the default of the correspondence-search / implicit-differentiation launches): split-f16
    operands, two-term layers calibrated once per weight version on the first call's points - value, Jacobian, Broyden
    roots and the conditioning gradient against the three-term launches of the same kernels."""
    net = U.build_deformation(device=dev).eval()
    for p in net.parameters():
        p.requires_grad_(False)
    mlp = net.defDeepSDF
    assert mlp.fit_numerics == "auto"
    g = torch.Generator().manual_seed(5)
    xyz = ((torch.rand(5, 1000, 3, generator=g) - 0.5) * 0.6).to(dev)
    cot = torch.randn(5, 1000, 3, generator=g).to(dev)
    lat0 = (torch.randn(5, 1, 1544, generator=g) * 0.05).to(dev)
    anc = (torch.from_numpy(U.anchors_mean()).float()[None] + 0.01 * torch.randn(5, 39, 3, generator=g)).to(dev)

    def run():
        lat = lat0.clone().requires_grad_(True)
        posed, J = net.posed_and_jacobian(xyz, lat, anc)
        (posed * cot).sum().backward()
        return posed.detach(), J, lat.grad

    p_a, J_a, g_a = run()
    rep = mlp._fit_cache
    # the criterion is value AND Jacobian on all five conditioning rows, in units of (fit_target, fit_jacobian_target)
    assert rep is not None and rep[2]["err"] <= 1.0 and rep[2]["rows"] == 5
    mlp.fit_numerics = "f16x3"
    p_x, J_x, g_x = run()
    e = (float((p_a - p_x).abs().max()), float((J_a - J_x).abs().max()), float((g_a - g_x).abs().max()) / float(g_x.abs().max()))
    print(f"fit tier, mask {rep[1]:#x} (sample {rep[2]['err']:.2e} of the bounds): posed {e[0]:.2e}, Jacobian {e[1]:.2e}, conditioning gradient (rel) {e[2]:.2e}")
    assert e[0] <= 2.0 * mlp.fit_target and e[1] <= 2.0 * mlp.fit_jacobian_target and e[2] < 5e-4
    # re-measurement on the current conditioning (what the fitting loop does every fit_verify_every steps): holds here;
    # with an unreachable bound the mask is withdrawn and the caller is told to record its graph again
    mlp.fit_numerics = "auto"
    run()
    assert mlp.reverify_fit() is False and mlp._fit_cache[1] == rep[1]
    keep = mlp.fit_target
    try:
        mlp.fit_target = 1e-12
        assert mlp.reverify_fit() is True and mlp._fit_cache[1] == 0
    finally:
        mlp.fit_target = keep
        mlp._fit_cache = None
    mlp.fit_numerics = "bf16x3"                        # rounds 1-3: still available, three-term on bf16 halves
    p_b, J_b, g_b = run()
    assert float((p_b - p_x).abs().max()) < 3e-6 and float((J_b - J_x).abs().max()) < 5e-5


@pytest.mark.parametrize("tier", ["f16x3", "auto"])
def test_value_jacobian_launch_cut_in_16_and_8_point_workgroups_is_bitwise_the_single_launch(dev, tier, monkeypatch):
    """The fitting batch (5 x 1000 points) runs its value+Jacobian launches as one chip-filling round of 16-point
    workgroups + the rest as 8-point workgroups (DeepSDF._jvp_split): values, Jacobians, and the conditioning gradient
    the saved state yields are bit-identical to the single 16-point launch (same per-point arithmetic, other geometry)."""
    net = U.build_deformation(device=dev).eval()
    net.defDeepSDF.fit_numerics = tier
    for p in net.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(33)
    R, n = 5, 1000
    xyz = ((torch.rand(R, n, 3, generator=g) - 0.5) * 0.6).to(dev)
    cot = torch.randn(R, n, 3, generator=g).to(dev)
    lat0 = (torch.randn(R, 1, 1544, generator=g) * 0.05).to(dev)
    anc = (torch.from_numpy(U.anchors_mean()).float()[None] + 0.01 * torch.randn(R, 39, 3, generator=g)).to(dev)
    mlp = net.defDeepSDF
    res = {}
    for split in ("1", "0"):
        monkeypatch.setenv("NPHM_AMD_JVP_SPLIT", split)
        cuts = mlp._jvp_split(R, n, dev, 64)
        # (the short round of 8-point workgroups is launched FIRST: a launch on another stream then finds free CUs at once)
        assert (len(cuts) == 2 and cuts[0][2] == 32 and cuts[1][2] == 64 and cuts[0][1] + cuts[1][1] == n and cuts[1][0] == 0
                and cuts[0][0] == cuts[1][1] and cuts[1][1] % 64 == 0) if split == "1" else cuts == [(0, 0, 64)]
        lat = lat0.clone().requires_grad_(True)
        # the saved-state buffer comes from the caching allocator unwritten past each row's end: hand it NaNs
        from nphm_amd import _lib
        nbytes = _lib.load().nphm_mlp_saved_bytes(*mlp._arch(), R, n)
        poison = torch.full((nbytes // 4,), float("nan"), device=dev)
        del poison
        posed, J = net.posed_and_jacobian(xyz, lat, anc)
        (posed * cot).sum().backward()
        assert bool(torch.isfinite(lat.grad).all())
        _, J2 = net.jacobian(xyz, lat0, anc)
        res[split] = (posed.detach().clone(), J.clone(), lat.grad.clone(), J2.clone())
    (p1, j1, g1, jj1), (p0, j0, g0, jj0) = res["1"], res["0"]
    assert torch.equal(p1, p0) and torch.equal(j1, j0) and torch.equal(jj1, jj0)
    # (the conditioning gradient: per-slot sums added in slot order - no atomics since ABI 8 - from the same saved state)
    assert torch.equal(g1, g0)
    # a launch too small or too large for the cut keeps the single form
    assert mlp._jvp_split(1, 1000, dev, 16) == [(0, 0, 64)] and mlp._jvp_split(64, 1000, dev, 16) == [(0, 0, 64)]


@pytest.mark.parametrize("hidden", [96, 400])
def test_k_loops_with_partial_last_round(dev, hidden, monkeypatch):
    """Hidden widths whose K-step count is not a multiple of four (96 -> 6, 400 -> 26): the four-slot register rings - the
    32-column value+Jacobian workgroups and the variant for all-two-term hidden layers - skip the slots of their last round
    that do not exist.  Value and Jacobian of the cut launch against the whole one (bitwise) and against autograd on the
    composite tier; the all-two-term lattice variant against the three-term kernel."""
    torch.manual_seed(hidden)
    net = nphm_amd.DeepSDF(lat_dim=29, hidden_dim=hidden, nlayers=4, geometric_init=False, out_dim=3).to(dev).eval()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.5)
    net.fit_numerics = "f16x3"
    g = torch.Generator().manual_seed(3)
    R, n = 5, 1000
    x = ((torch.rand(R, n, 3, generator=g) - 0.5) * 0.8).to(dev)
    cond = (torch.randn(R, 29, generator=g) * 0.3).to(dev)
    outs = {}
    for split in ("1", "0"):
        monkeypatch.setenv("NPHM_AMD_JVP_SPLIT", split)
        assert len(net._jvp_split(R, n, dev, 16)) == (2 if split == "1" else 1)
        outs[split] = net.forward_hip_jvp(x, cond)
    assert torch.equal(outs["1"], outs["0"])
    net.backend = "composite"
    xg = x.clone().requires_grad_(True)
    val, _ = net(xg, cond[:, None, :].expand(R, n, 29))
    jac = torch.stack([torch.autograd.grad(val[..., i].sum(), xg, retain_graph=True)[0] for i in range(3)], dim=-1)   # [R,n,3(x),3(out)]
    net.backend = "hip"
    scale = float(jac.abs().max())
    assert U.maxdiff(outs["1"][:, :, 0].cpu().numpy(), val.detach().cpu().numpy()) < 2e-5 * max(1.0, float(val.detach().abs().max()))
    assert U.maxdiff(outs["1"][:, :, 1:].cpu().numpy(), jac.cpu().numpy()) < 5e-5 * max(1.0, scale)
    # lattice-sized evaluation: every hidden layer two-term (the ALL2 variant) against three terms everywhere
    big = ((torch.rand(1, 1 << 18, 3, generator=g) - 0.5) * 0.8).to(dev)
    net.numerics, net.two_pass_mask = "fixed", 0
    ref = net.forward_hip(big, cond[:1])
    net.two_pass_mask = (1 << 4) - 2                                   # hidden GEMM layers 1 .. 3 of the 5 linear layers
    two = net.forward_hip(big, cond[:1])
    err = float((two - ref).abs().max())
    assert 0 < err < 2e-4 * max(1.0, float(ref.abs().max())), err


@pytest.mark.parametrize("hidden,nlayers,lat,out", [(64, 2, 5, 1), (96, 4, 29, 3), (400, 6, 40, 4), (512, 3, 8, 2), (1024, 4, 64, 1), (640, 5, 16, 3)])
def test_tiers_and_fp32_last_layer_across_architectures(dev, hidden, nlayers, lat, out):
    """Every architecture family the kernel covers (hidden <= 512: 64 / 128 points per workgroup, hidden <= 1024: 32 / 64;
    2 .. 6 layers, 1 .. 4 outputs, partial tiles and partial K rounds) in every tier - three-term, two-term, single-term on all
    hidden layers (the variant without the lo plane), that with the last hidden layer two-term, and a mixed mask - against the composite tier in fp32 autograd
    arithmetic, points launch with a ragged tail and lattice launch (bitwise the points launch)."""
    torch.manual_seed(hidden + nlayers)
    net = nphm_amd.DeepSDF(lat_dim=lat, hidden_dim=hidden, nlayers=nlayers, geometric_init=False, out_dim=out).to(dev).eval()
    if not net.hip_supported():
        pytest.skip("architecture outside the kernel's plan")
    g = torch.Generator().manual_seed(7)
    x = ((torch.rand(1, 1000 + 77, 3, generator=g) - 0.5) * 0.9).to(dev)
    cond = (torch.randn(1, lat, generator=g) * 0.3).to(dev)
    net.backend = "composite"
    with torch.no_grad():
        ref, _ = net(x, cond[:, None, :].expand(1, x.shape[1], lat))
    net.backend = "hip"
    hid = net._hidden_mask()
    mixed_one = hid & 0b101010
    scale = max(1.0, float(ref.abs().max()))
    net.numerics = "fixed"
    axes = R.grid_axes(U.MINI, U.MAXI, 12)
    pts = torch.from_numpy(np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)).to(dev)
    tail = 1 << (nlayers - 1)                                # the last hidden layer (two point halves in the variant without a lo plane)
    for name, two, one, tol in (("three", 0, 0, 2e-6), ("two", hid, 0, 5e-5), ("single", 0, hid, 2e-4), ("mixed", hid, mixed_one, 2e-4),
                                ("single, last two-term", tail, hid & ~tail, 2e-4)):
        net.two_pass_mask, net.single_mask = two, one
        with torch.no_grad():
            outp = net.forward_hip(x, cond)
            err = float((outp - ref).abs().max()) / scale
            vol = R.evaluate_grid_mlp(net, cond, axes)
            assert torch.equal(vol.reshape(-1, out), net.forward_hip(pts, cond).reshape(-1, out)), name
        print(f"hidden {hidden} x {nlayers} layers, out {out}, {name}: {err:.2e} of the output scale against fp32 autograd")
        assert err < tol, (name, err)


@pytest.mark.parametrize("hidden,nlayers,lat,out", [(1024, 8, 64, 1), (640, 5, 16, 3), (512, 3, 8, 2), (400, 6, 40, 4), (96, 4, 29, 3)])
def test_last_hidden_layer_in_k_halves_is_bitwise_the_two_point_halves_form(dev, hidden, nlayers, lat, out):
    """The tier set "single-term everywhere, last hidden layer two-term" (NPM's): with a workspace the last hidden layer runs
    in two K halves - the second half of its operands parked in a slot of the workspace, back by LDS-DMA - and streams its
    weights once; without one, in two point halves.  Same products in the same order per accumulator: the same bits.  Sizes:
    more workgroups than slots (the slots are taken and released), a ragged tail, two conditioning rows, partial K halves
    (640: 20 tiles = 16 + 4; 400, 96: the second half is short or empty), and the lattice launch."""
    torch.manual_seed(3 * hidden + nlayers)
    net = nphm_amd.DeepSDF(lat_dim=lat, hidden_dim=hidden, nlayers=nlayers, geometric_init=False, out_dim=out).to(dev).eval()
    assert net.hip_supported()
    g = torch.Generator().manual_seed(11)
    n = 64 * 1024 + 37 if hidden > 512 else 128 * 700 + 5
    x = ((torch.rand(2, n, 3, generator=g) - 0.5) * 0.9).to(dev)
    cond = (torch.randn(2, lat, generator=g) * 0.3).to(dev)
    tail = 1 << (nlayers - 1)
    net.numerics, net.two_pass_mask, net.single_mask = "fixed", tail, net._hidden_mask() & ~tail
    lib = nphm_amd._lib.load()
    code = int(net._format_code() | (tail << 8) | ((net._hidden_mask() & ~tail) << 20))
    packed, state = net.prepare_latent(cond)
    calls = {"ws": 0, "plain": 0}
    orig = lib.nphm_mlp_eval_points_ws

    def spy(*a):
        calls["ws" if a[-3] else "plain"] += 1
        return orig(*a)

    lib.nphm_mlp_eval_points_ws = spy
    try:
        with torch.no_grad():
            net.tail_k_split = True
            a = net._eval_points_raw(packed, state, x, False, code)
            a2 = net._eval_points_raw(packed, state, x, False, code)      # (again: the slots of the first launch were all released)
            net.tail_k_split = False
            b = net._eval_points_raw(packed, state, x, False, code)
            one_row = net.forward_hip(x[:1], cond[:1])                    # the module's own route to the same tiers ("fixed")
    finally:
        lib.nphm_mlp_eval_points_ws = orig
    assert torch.equal(one_row, a[:1])
    assert calls == {"ws": 2, "plain": 2}
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0
    assert torch.equal(a, b) and torch.equal(a, a2)
    axes = R.grid_axes(U.MINI, U.MAXI, 40)
    with torch.no_grad():
        net.tail_k_split = True
        va = R.evaluate_grid_mlp(net, cond[:1], axes)
        net.tail_k_split = False
        vb = R.evaluate_grid_mlp(net, cond[:1], axes)
    assert torch.equal(va, vb)
    # a workspace that is too small is refused, not overrun
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    o = torch.empty(1, 64, out, device=dev)
    rc = lib.nphm_mlp_eval_points_ws(*net._arch(), packed.data_ptr(), state.data_ptr(), x.data_ptr(), 1, 64, 0, code, o.data_ptr(),
                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    assert rc != 0
