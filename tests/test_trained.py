"""Parity on a TRAINED-LIKE checkpoint (tests/golden/trained_state.npz: the identity decoder after 5 000 training
steps on analytic head-like surfaces - weights up to 1.25 against 0.07 at the seeded init, |sdf| up to 0.35 in the
extraction box, a real zero level set).  tests/golden/trained.npz holds the REFERENCE's outputs on that state_dict
(make_golden_trained.py: FastEnsembleDeepSDFMirrored of /root/reference on the CPU, strict load).

CPU tests pin the numpy oracle and the composite tier to those outputs; GPU tests pin every precision mode of the HIP
kernels - the default (adaptive split-bf16, prune_tol 1e-7) at a tenth of the 1e-4 bar - and report the guard's
verdict (nphm_amd.validate_numerics) and the member statistics of the checkpoint."""
import os

import numpy as np
import pytest
import torch

import _util as U
import nphm_amd
from nphm_amd import reconstruction as R
from oracle import nphm_oracle as O

TOL_BAR = 1e-4
CODES = (0, 1, 2, 17)


@pytest.fixture(scope="module")
def fx():
    return U.golden("trained")


def _voxel_points(fx, c, res=256):
    axes = R.grid_axes(U.MINI, U.MAXI, res)
    keep = fx[f"c{c}_voxels"]
    return np.stack([axes[0][keep // (res * res)], axes[1][(keep // res) % res], axes[2][keep % res]], -1).astype(np.float32)


def test_checkpoint_is_the_one_the_reference_evaluated(fx):
    net, codes = U.build_trained_identity()
    assert U.state_hash(net) == str(fx["state_sha256"])
    assert codes.shape == (64, 1344) and float(codes.norm(dim=-1).max()) <= 1.0 + 1e-5
    w = net.state_dict()["ensembled_deep_sdf.lin2.weight"]
    assert float(w.abs().max()) > 1.0          # trained sharpness: the seeded init is bounded by 1 / sqrt(200)


def test_oracle_and_composite_match_the_reference_on_trained_weights(fx):
    net, codes = U.build_trained_identity()
    net.backend = "composite"
    net.train()
    params, amean = U.np_state(net), U.anchors_mean()
    for c in CODES[:2]:
        lat = codes[c].numpy()[None, None]
        pts = np.concatenate([_voxel_points(fx, c)[::6], fx[f"c{c}_near"][::4]])[None]        # 768 + 512 points
        ref = np.concatenate([fx[f"c{c}_sdf_voxels"][::6], fx[f"c{c}_sdf_near"][::4]])
        got_o, anc_o = O.nphm_identity_forward(params, amean, pts, lat, training=True)
        assert U.maxdiff(anc_o[0], fx[f"c{c}_anchors"]) < 1e-6
        assert U.maxdiff(got_o.reshape(-1), ref) < 2e-6
        with torch.no_grad():
            got_c, _ = net(torch.from_numpy(pts), codes[c][None, None], None)
        assert U.maxdiff(got_c.reshape(-1).numpy(), ref) < 2e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("precision,prune,tol", [
    ("auto", None, 1e-5),                                   # the default: knobs calibrated for THIS checkpoint, a decade inside the bar
    # exact / split products, all 40 members: measured 4.0e-7 / 1.1e-6 / 3.7e-7 (round 3), asserted at ~2x
    ("f32", -1.0, 1e-6), ("bf16x3", -1.0, 2.5e-6), ("f16x3", -1.0, 1e-6),
    # plain pruning budgets: measured 1.4e-5 at 1e-7 (a trained member predicts tens of SDF units where its weight is 1e-7)
    ("f16x3", 1e-8, 1e-5), ("bf16x3", 1e-7, 3e-5), ("f16x3", 1e-7, 3e-5),
    # PINNED tiered modes at their seeded-weight thresholds: 2.9e-5 (bf16x3a2) .. 2.1e-4 (f16x3a2, outside the bar) when run as
    # pinned in round 3 - the guard of pinned approximate modes (numerics.clamp_pinned, 2e-5 on a sample) now re-tiers them
    ("bf16x3a", 1e-7, 4e-5), ("bf16x3a2", 1e-7, 4e-5), ("f16x3a2", 1e-7, 4e-5)])
def test_hip_modes_on_trained_weights(fx, dev, precision, prune, tol):
    """every precision mode against the reference fixture: 4 codes x (4 608 stratified voxels of a 256^3 extraction + 2 048
    near-surface points)"""
    net, codes = U.build_trained_identity(device=dev)
    net.train()                                  # the fixture's point sets are train-mode values (no last-point overwrite)
    if precision != "auto":
        net.precision, net.prune_tol = precision, prune
    worst = {}
    axes = R.grid_axes(U.MINI, U.MAXI, 256)
    for c in CODES:
        with torch.no_grad():
            # the stratified voxels out of a REAL 256^3 extraction (compact 4x4x2 tiles per wavefront: pruning and the
            # precision tiers are decided per wavefront, scattered points would share more members and err less)
            vol = R.evaluate_grid(net, codes[c], axes, hack_chunk=0)
            err = np.abs(vol[torch.from_numpy(fx[f"c{c}_voxels"]).to(dev)].cpu().numpy() - fx[f"c{c}_sdf_voxels"])
            for i, reg in enumerate(("near_anchor", "mid_field", "far_field")):
                worst[reg] = max(worst.get(reg, 0.0), float(err[1536 * i:1536 * (i + 1)].max()))
            got, anc = net(torch.from_numpy(fx[f"c{c}_near"])[None].to(dev), codes[c][None, None], None)
            worst["near_surface"] = max(worst.get("near_surface", 0.0),
                                        float(np.abs(got.reshape(-1).cpu().numpy() - fx[f"c{c}_sdf_near"]).max()))
            assert U.maxdiff(anc[0].cpu().numpy(), fx[f"c{c}_anchors"]) < 1e-6
    if precision == "auto":
        c = net.calibration
        print(f"trained checkpoint, calibrated: {c['precision']} light {c['light_tol']} mid {c['mid_tol']} prune {c['prune_tol']:g} "
              f"(sample error {c['error']:.2e})")
        prune = c["prune_tol"]
    elif net._pinned_check is not None:
        r = net._pinned_check[1]
        print(f"pinned guard: asked {r['asked']} -> error {r['asked_error']:.2e}; runs {({k: r[k] for k in r['asked']})} -> {r['error']:.2e}"
              f" (clamped: {r['clamped']})")
        if precision == "f16x3a2":
            # the mode round 3 shipped at 2.1e-4 on these voxels (8.5e-5 on the guard's 512-tile sample): re-tiered
            assert r["clamped"] and r["asked_error"] > r["guard"] >= r["error"]
    print(f"trained checkpoint, {precision}, prune_tol {prune:g}: max |hip - reference| " +
          ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert max(worst.values()) < tol


@pytest.mark.gpu
def test_get_logits_eval_mode_lattice_on_trained_weights(fx, dev):
    """the reference's get_logits (eval mode, chunk 9 000: overwrite voxels included) on a 40^3 lattice"""
    net, codes = U.build_trained_identity(device=dev)
    net.eval()
    res, chunk = int(fx["lattice_res"]), int(fx["lattice_chunk"])
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None].to(dev)
    vol = R.get_logits(net, codes[CODES[0]], grid, nbatch_points=chunk)
    ref = fx["lattice_volume"]
    hacked = O.hack_indices(res ** 3, chunk)
    assert U.maxdiff(vol[hacked], ref[hacked]) < 1e-6          # S / (S + 1e-6): 1 near the head, ~0 in the far field
    err = np.abs(vol - ref)
    print(f"trained checkpoint, 40^3 get_logits: max |err| {err.max():.2e}, sign flips {int(((vol < 0) != (ref < 0)).sum())}")
    assert float(err.max()) < 1e-5


@pytest.mark.gpu
def test_validate_numerics_and_member_statistics_on_trained_weights(dev):
    net, codes = U.build_trained_identity(device=dev)
    net.eval()
    rep = nphm_amd.validate_numerics(net, codes[list(CODES)], n=1 << 17)         # numerics = "auto": the calibrated setting
    print("validate_numerics on the trained checkpoint:", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in rep.items()})
    assert rep["ok"] and rep["max_abs_diff"] < 1e-5 and rep["max_abs_sdf"] > 0.2
    net.precision, net.prune_tol = "bf16x3a2", 1e-7                              # the pinned round-2 default is flagged
    rep2 = nphm_amd.validate_numerics(net, codes[list(CODES)], n=1 << 17)
    print("... pinned bf16x3a2 / 1e-7:", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in rep2.items()})
    assert not rep2["ok"] and rep2["max_abs_diff"] < TOL_BAR
    net.numerics = "auto"
    # member statistics of the default mode on the 128^3 lattice (what the throughput depends on)
    axes = R.grid_axes(U.MINI, U.MAXI, 128)
    stats = torch.zeros(16, dtype=torch.int64, device=dev)
    R.evaluate_grid(net, codes[0], axes, hack_chunk=0, stats=stats)          # calibrated setting
    s = stats.cpu().numpy()
    n = 128 ** 3
    print(f"trained checkpoint 128^3: members per wavefront {s[0] / n:.2f} (single-pass {s[15] / n:.2f}, two-pass {s[14] / n:.2f})")
    assert 2.0 < s[0] / n < 20.0


@pytest.mark.gpu
def test_fitting_tier_on_trained_weights(dev):
    """The first-order autograd tier (what the latent fitting loops differentiate) on the trained-like checkpoint: values
    and the gradients w.r.t. points and code against the composite PyTorch tier at 5 000 near-surface points.  With
    numerics = "auto" the pruning budget follows the size of the member values (measured once per weight version): the
    plain 1e-7, calibrated on the seeded initialisation, leaves 1.7e-5 / 3.7e-4 here and is shown next to it."""
    net, codes = U.build_trained_identity(device=dev)
    net.train()
    for p in net.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    surf = torch.from_numpy(np.load(os.path.join(U.GOLDEN, "trained_state.npz"))["subject_anchors"]).float()
    c = 2
    i, j = torch.randint(0, 39, (5000,), generator=g), torch.randint(0, 39, (5000,), generator=g)
    t = torch.rand(5000, 1, generator=g) * 0.3
    x0 = (surf[c][i] * (1 - t) + surf[c][j] * t + 0.01 * torch.randn(5000, 3, generator=g)).to(dev)[None]
    seed = torch.randn(1, 5000, 1, generator=g).to(dev)

    def run(backend):
        net.backend = backend
        lat = codes[c][None, None].clone().requires_grad_()
        x = x0.clone().requires_grad_()
        sdf, _ = net(x, lat, None)
        (sdf * seed).sum().backward()
        return sdf.detach(), x.grad.clone(), lat.grad.clone()

    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    ref = run("composite")
    assert net.numerics == "auto"
    out = run("hip")
    e_auto = (float((out[0] - ref[0]).abs().max()), rel(out[1], ref[1]), rel(out[2], ref[2]))
    net.prune_tol = 1e-7                                   # pinned: the plain rule
    out = run("hip")
    e_plain = (float((out[0] - ref[0]).abs().max()), rel(out[1], ref[1]), rel(out[2], ref[2]))
    print(f"fitting tier on the trained-like checkpoint: auto budget |sdf| {e_auto[0]:.1e}, d/dx {e_auto[1]:.1e}, d/dcode {e_auto[2]:.1e}; "
          f"plain 1e-7: {e_plain[0]:.1e}, {e_plain[1]:.1e}, {e_plain[2]:.1e}")
    assert e_auto[0] < 1e-5 and e_auto[1] < 3e-4 and e_auto[2] < 1e-5
    assert e_plain[0] < TOL_BAR and e_plain[0] > e_auto[0]


@pytest.mark.gpu
def test_mesh_of_a_trained_code_matches_the_composite_tier(dev):
    """the north star's mesh criterion on trained-like weights: get_logits -> mesh_from_logits at 96^3 on the calibrated HIP
    path and on the composite PyTorch tier (the reference's arithmetic), Chamfer distance of the two meshes < 1e-5"""
    from scipy.spatial import cKDTree
    from NPHM.utils.reconstruction import create_grid_points_from_bounds, mesh_from_logits
    net, codes = U.build_trained_identity(device=dev)
    net.eval()
    res = 96
    grid = torch.from_numpy(create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None].to(dev)
    vol_h = R.get_logits(net, codes[1], grid, nbatch_points=25000)
    net.backend = "composite"
    vol_c = R.get_logits(net, codes[1], grid, nbatch_points=25000)
    mh, mc = mesh_from_logits(vol_h, U.MINI, U.MAXI, res), mesh_from_logits(vol_c, U.MINI, U.MAXI, res)
    vh, vc = np.asarray(mh.vertices), np.asarray(mc.vertices)
    chamfer = max(cKDTree(vc).query(vh)[0].mean(), cKDTree(vh).query(vc)[0].mean())
    print(f"trained code 1, 96^3: {len(vh)} / {len(vc)} vertices, Chamfer {chamfer:.2e}, max |volume difference| {np.abs(vol_h - vol_c).max():.2e}")
    assert len(vc) > 5000 and abs(len(vh) - len(vc)) <= 0.002 * len(vc)
    assert chamfer < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("weights", ["seeded", "trained"])
def test_auto_numerics_holds_for_latents_it_was_not_calibrated_with(dev, weights):
    """numerics = "auto" calibrates with the FIRST large evaluation's latent; every later latent's knobs are verified on a sample
    of that latent (and re-calibrated on the union when the sample disagrees).  Eight other latents - up to 3 sigma / up to
    twice as far from the mean code as any training code - as full 128^3 volumes against the dense exact-fp32 kernel."""
    if weights == "seeded":
        net = U.build_identity(device=dev).eval()
        first = U.sample_latent(0).to(dev)
        others = [U.sample_latent(10 + i, scale=sc).to(dev) for i, sc in enumerate((0.85, 0.85, 1.5, 1.5, 1.5, 3.0, 3.0, 3.0))]
    else:
        net, codes = U.build_trained_identity(device=dev)
        net.eval()
        first = codes[0]
        mean = codes.mean(0)
        others = [codes[i] for i in (1, 2, 3, 5, 8, 13)] + [mean + 2.0 * (codes[21] - mean), mean + 2.0 * (codes[34] - mean)]
    axes = R.grid_axes(U.MINI, U.MAXI, 128)
    with torch.no_grad():
        R.evaluate_grid(net, first, axes, hack_chunk=0)                        # calibrates (and verifies) with `first`
        cal0 = dict(net.calibration)
        worst, recal = 0.0, 0
        for lat in others:
            fast = R.evaluate_grid(net, lat, axes, hack_chunk=0)
            assert net.numerics == "auto"
            knobs = (net.calibration["precision"], net.calibration["light_tol"], net.calibration["mid_tol"], net.calibration["prune_tol"])
            recal += int("recalibrated_for" in net.calibration)
            net.precision, net.prune_tol = "f32", -1.0                         # dense exact-fp32 reference (pins the numerics)
            ref = R.evaluate_grid(net, lat, axes, hack_chunk=0)
            net.numerics = "auto"
            e = float((fast - ref).abs().max())
            worst = max(worst, e)
            print(f"{weights}: latent |z - first| {float((lat - first).norm()):.2f}, knobs {knobs}: max |auto - dense f32| = {e:.2e}")
        # ... and against the numpy ORACLE (not a kernel of this repo) on a stratified subsample of the last volume: voxels next
        # to the anchors, at mid range and far out, the strata test_full_size_256_cubed_properties uses (verdict round 5, item 3)
        anchors = net.prepare_latent(lat[None])[2][0].cpu().numpy()
        idx = U.stratified_voxels(axes, anchors, 700, seed=5)
        ax, ay, az = [np.asarray(a, np.float32) for a in axes]
        ix, iy, iz = np.unravel_index(idx, (len(ax), len(ay), len(az)))
        pts = np.stack([ax[ix], ay[iy], az[iz]], axis=-1)[None].astype(np.float32)
        lat_rep = np.repeat(lat.cpu().numpy()[None, None], pts.shape[1], axis=1)
        want, _ = O.nphm_identity_forward(U.np_state(net), U.anchors_mean(), pts, lat_rep, training=True)
        e_oracle = U.maxdiff(fast.cpu().numpy()[idx], np.asarray(want).reshape(-1))
        print(f"{weights}: last latent, {len(idx)} stratified voxels against the numpy oracle: max |auto - oracle| = {e_oracle:.2e}")
        assert e_oracle <= 1e-5
    print(f"{weights}: calibrated with the first latent as {cal0['precision']} / {cal0['light_tol']} / {cal0['mid_tol']} / "
          f"{cal0['prune_tol']:g}; worst later latent {worst:.2e}; re-calibrations {recal}; verified latents {len(net._verified_latents)}")
    assert worst <= 1e-5
    assert len(net._verified_latents) >= 1 + len(others) - recal


@pytest.mark.gpu
def test_small_evaluations_and_captured_streams_stay_exact(dev):
    """Below AUTO_MIN_POINTS points numerics = "auto" runs every member on the three-pass product (nothing to calibrate, nothing
    to trust); a captured stream cannot synchronise, so it never calibrates or verifies there."""
    net, codes = U.build_trained_identity(device=dev)
    net.eval()
    pts = (torch.rand(1, 4096, 3, device=dev) - 0.5)
    with torch.no_grad():
        a, _ = net(pts, codes[3][None, None], None)
        assert net.calibration is None                                          # no calibration was needed
        net.precision, net.prune_tol = "f16x3", -1.0
        b, _ = net(pts, codes[3][None, None], None)
        assert torch.equal(a, b)
        net.numerics = "auto"
        g = torch.cuda.CUDAGraph()
        big = (torch.rand(1, 1 << 17, 3, device=dev) - 0.5)
        out = torch.empty(1, 1 << 17, 1, device=dev)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net(big[:, :64], codes[4][None, None], None)                        # warm-up outside the capture (packs the weights)
            torch.cuda.current_stream().synchronize()
            with torch.cuda.graph(g, stream=s):
                out.copy_(net(big, codes[4][None, None], None)[0])
        g.replay()
        torch.cuda.synchronize()
        assert net.calibration is None                                          # captured: exact setting, no calibration
        net.precision, net.prune_tol = "f16x3", -1.0
        ref, _ = net(big, codes[4][None, None], None)
        assert torch.equal(out, ref)


# ---- round 4: the trained-like DEFORMATION checkpoint (reference module trained on analytic expression warps) -------------------
def _def_fixture():
    return U.golden("trained_def")


def test_oracle_and_composite_tier_reproduce_the_trained_deformation_fixture():
    """CPU: the numpy oracle and this repo's composite tier against the reference's outputs on the trained deformation network"""
    fx = _def_fixture()
    dnet, z_ex, pairs = U.build_trained_deformation()
    assert U.state_hash(dnet) == str(fx["state_sha256"])
    _, codes = U.trained_checkpoint()
    dnet.defDeepSDF.backend = "composite"
    params = U.np_state(dnet)
    for i, (s, e) in enumerate(pairs):
        lat = torch.cat([codes[s], z_ex[i]])[None, None]
        x, anc = torch.from_numpy(fx[f"p{i}_xyz"]), torch.from_numpy(fx[f"p{i}_anchors"])
        with torch.no_grad():
            off, _ = dnet(x, lat, anc)
        assert U.maxdiff(off.numpy(), fx[f"p{i}_offsets"]) < 2e-6
        off_o, _ = O.deformation_forward(params, fx[f"p{i}_xyz"], lat.numpy(), fx[f"p{i}_anchors"])
        assert U.maxdiff(off_o, fx[f"p{i}_offsets"]) < 2e-6
    assert float(np.abs(fx["p0_offsets"]).max()) > 5e-3 and float(fx["max_weight"]) > 0.2      # it IS a trained network


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("f16x3", 1e-6), ("bf16x3", 5e-6)])
def test_hip_deformation_on_trained_weights(dev, precision, tol):
    fx = _def_fixture()
    dnet, z_ex, pairs = U.build_trained_deformation(device=dev)
    _, codes = U.trained_checkpoint()
    codes = codes.to(dev)
    dnet.defDeepSDF.precision, dnet.defDeepSDF.numerics = precision, "fixed"
    worst = 0.0
    for i, (s, e) in enumerate(pairs):
        lat = torch.cat([codes[s], z_ex[i]])[None, None]
        with torch.no_grad():
            off, _ = dnet(torch.from_numpy(fx[f"p{i}_xyz"]).to(dev), lat, torch.from_numpy(fx[f"p{i}_anchors"]).to(dev))
        worst = max(worst, U.maxdiff(off.cpu().numpy(), fx[f"p{i}_offsets"]))
    print(f"trained deformation checkpoint, {precision}: max |hip - reference| = {worst:.2e} (offsets up to {float(np.abs(fx['p0_offsets']).max()):.2e})")
    assert worst < tol


@pytest.mark.gpu
def test_two_stage_on_trained_identity_and_deformation_weights(dev):
    """the reference's get_logits_backward (eval mode, chunk 1 500: overwrite voxels included) on a 20^3 lattice, both networks
    trained-like; and the two-term tier of the deformation stage on THIS checkpoint (what `auto` decides, and what it costs)"""
    fx = _def_fixture()
    dnet, z_ex, pairs = U.build_trained_deformation(device=dev)
    inet, codes = U.build_trained_identity(device=dev)
    inet.eval()
    s, e = pairs[0]
    res, chunk = int(fx["lattice_res"]), int(fx["lattice_chunk"])
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None].to(dev)
    lat_all = torch.cat([codes[s], z_ex[0]])
    vol = R.get_logits_backward(inet, dnet, codes[s], lat_all, grid, nbatch_points=chunk,
                                anchors=torch.from_numpy(fx["p0_anchors"]).to(dev))
    err = np.abs(vol - fx["two_stage_logits"])
    print(f"two-stage, trained identity + trained deformation, {res}^3: max |err| {err.max():.2e}")
    assert float(err.max()) < 1e-5
    # a lattice-sized evaluation of the deformation stage: the calibrated two-term layers against the three-term product
    mlp = dnet.defDeepSDF
    gen = torch.Generator().manual_seed(2)
    xyz = ((torch.rand(1, 1 << 18, 3, generator=gen) - 0.5) * 0.9).to(dev)
    _, cond = R._expr_condition(dnet, lat_all, torch.from_numpy(fx["p0_anchors"]).to(dev), dev)
    with torch.no_grad():
        mlp.numerics, mlp.two_pass_mask = "fixed", 0
        ref = mlp.forward_hip(xyz, cond.reshape(1, -1))
        mlp.numerics = "auto"
        auto = mlp.forward_hip(xyz, cond.reshape(1, -1))
    rep = mlp.last_numerics
    e_auto = float((auto - ref).abs().max())
    print(f"trained deformation checkpoint: auto mask {rep['mask']:#x} (all-layers error {rep['all_layers_err']:.2e}, "
          f"per layer {rep.get('per_layer_err')}), full-set error {e_auto:.2e}, outputs up to {float(ref.abs().max()):.2e}")
    assert e_auto <= 2.0 * mlp.two_pass_target


# ---- round 4: the trained-like NPM checkpoint (reference DeepSDF trained on analytic head surfaces) -------------------------
def test_oracle_and_composite_tier_reproduce_the_trained_npm_fixture():
    fx = U.golden("trained_npm")
    net, codes = U.build_trained_npm()
    assert U.state_hash(net) == str(fx["state_sha256"])
    net.backend = "composite"
    params = U.np_state(net)
    for i in range(codes.shape[0]):
        x = torch.from_numpy(fx[f"c{i}_xyz"])
        with torch.no_grad():
            sdf, _ = net(x, codes[i][None, None])
        assert U.maxdiff(sdf.numpy(), fx[f"c{i}_sdf"]) < 5e-6
        lat_rep = np.repeat(codes[i].numpy()[None, None], x.shape[1], axis=1)
        sdf_o = O.deepsdf_forward(params, "", fx[f"c{i}_xyz"], lat_rep, nlayers=8)
        assert U.maxdiff(np.asarray(sdf_o).reshape(fx[f"c{i}_sdf"].shape), fx[f"c{i}_sdf"]) < 5e-6
    assert float(fx["max_weight"]) > 0.15


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("f16x3", 3e-6), ("bf16x3", 1e-5)])
def test_hip_npm_on_trained_weights(dev, precision, tol):
    fx = U.golden("trained_npm")
    net, codes = U.build_trained_npm(device=dev)
    net.precision, net.numerics = precision, "fixed"
    worst = 0.0
    for i in range(codes.shape[0]):
        with torch.no_grad():
            sdf, _ = net(torch.from_numpy(fx[f"c{i}_xyz"]).to(dev), codes[i][None, None])
        worst = max(worst, U.maxdiff(sdf.cpu().numpy(), fx[f"c{i}_sdf"]))
    res, chunk = int(fx["lattice_res"]), int(fx["lattice_chunk"])
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None].to(dev)
    vol = R.get_logits(net, codes[0], grid, nbatch_points=chunk)
    e_lat = U.maxdiff(vol, fx["lattice_logits"])
    print(f"trained NPM checkpoint, {precision}: points {worst:.2e}, {res}^3 get_logits {e_lat:.2e}")
    assert worst < tol and e_lat < tol


@pytest.mark.gpu
def test_two_term_tier_on_trained_npm_weights(dev):
    net, codes = U.build_trained_npm(device=dev)
    axes = R.grid_axes(U.MINI, U.MAXI, 64)
    with torch.no_grad():
        net.numerics, net.two_pass_mask = "fixed", 0
        ref = R.evaluate_grid_mlp(net, codes[1][None], axes)
        net.numerics = "auto"
        auto = R.evaluate_grid_mlp(net, codes[1][None], axes)
    rep = net.last_numerics
    e = float((auto - ref).abs().max())
    print(f"trained NPM checkpoint, 64^3: auto mask {rep['mask']:#x} (all-layers error {rep['all_layers_err']:.2e}), full-volume error {e:.2e}")
    assert e <= 2.0 * net.two_pass_target


@pytest.mark.gpu
def test_auto_defers_calibration_while_the_weights_churn(dev):
    """A training loop that extracts one validation volume per weight version (training.py:312-323) must not pay a
    calibration (~0.17 s, worth it after ~10 volumes) for every version: when a calibration has served fewer than CHURN_USES
    large evaluations before the weights change, the NEW weights run the exact three-pass setting until a second large
    evaluation sees them unchanged."""
    net, codes = U.build_trained_identity(device=dev)
    net.eval()
    axes = R.grid_axes(U.MINI, U.MAXI, 48)
    lat = codes[2]
    exact = (net._EXACT_KNOBS[0], net.precision_code(net._EXACT_KNOBS[1]))
    with torch.no_grad():
        R.evaluate_grid(net, lat, axes, hack_chunk=0)                 # first weights of the process: calibrated at once
        key0 = net._calibration[0]
        assert net.calibration is not None and net._auto_hist["uses"] >= 1
        net.ensembled_deep_sdf.lin4.bias.add_(1e-4)                   # "an optimizer step": a new weight version
        n = 48 ** 3
        assert net.kernel_knobs(dev, lat[None], n) == exact and net._calibration[0] == key0          # deferred: exact setting
        vol_exact = R.evaluate_grid(net, lat, axes, hack_chunk=0)    # second sight of the same weights: calibrates now
        assert net._calibration[0] != key0
        net.precision, net.prune_tol = "f16x3", -1.0
        ref = R.evaluate_grid(net, lat, axes, hack_chunk=0)
        assert float((vol_exact - ref).abs().max()) <= 1e-5
        # a well-used calibration: the next weight version is calibrated immediately
        net.numerics = "auto"
        for _ in range(net.CHURN_USES):
            R.evaluate_grid(net, lat, axes, hack_chunk=0)
        key1 = net._calibration[0]
        net.ensembled_deep_sdf.lin4.bias.add_(1e-4)
        R.evaluate_grid(net, lat, axes, hack_chunk=0)
        assert net._calibration[0] != key1


# ---- round 6: the calibrated tiers against the REFERENCE at lattice size (tests/golden/make_golden_trained_lattice.py) ----------
# The small trained fixtures (<= 8 000 points) sit below the sizes at which numerics = "auto" leaves the three-term product
# (DeepSDF.two_pass_min_points = 262 144, AUTO_MIN_POINTS = 65 536): there the tiers were only compared with this repo's own
# three-term kernel.  These four evaluate the 64^3 lattice the reference's modules were run on (262 144 points) with `auto`.
def _lattice_fixture():
    fx = U.golden("trained_lattice")
    res, chunk = int(fx["res"]), int(fx["chunk"])
    return fx, res, chunk, R.grid_axes(U.MINI, U.MAXI, res)


@pytest.mark.gpu
def test_auto_tiers_of_the_deformation_net_match_the_reference_at_lattice_size(dev):
    fx, res, chunk, axes = _lattice_fixture()
    dnet, z_ex, pairs = U.build_trained_deformation(device=dev)
    assert U.state_hash(dnet) == str(fx["deformation_sha256"])
    _, codes = U.build_trained_identity(device=dev)
    i = pairs.index(tuple(int(v) for v in fx["pair"]))
    lat_all = torch.cat([codes[pairs[i][0]], z_ex[i]])
    mlp, cond = R._expr_condition(dnet, lat_all, torch.from_numpy(fx["anchors"]).to(dev), dev)
    assert mlp.numerics == "auto" and res ** 3 >= mlp.two_pass_min_points
    with torch.no_grad():
        off = R.evaluate_grid_mlp(mlp, cond, axes)
    rep = mlp.last_numerics
    e = U.maxdiff(off.cpu().numpy(), fx["def_offsets"])
    print(f"trained deformation net, {res}^3, auto tiers (single-term layers {rep.get('single_mask', 0):#x}, two-term {rep.get('mask', 0):#x}, "
          f"sample error {rep.get('verified_err', 0.0):.2e}): max |hip - reference| = {e:.2e} (offsets up to {np.abs(fx['def_offsets']).max():.2e})")
    assert rep.get("single_mask", 0) != 0 or rep.get("mask", 0) != 0          # a calibrated tier did run
    assert e <= 1e-5


@pytest.mark.gpu
def test_auto_tiers_of_the_npm_net_match_the_reference_at_lattice_size(dev):
    fx, res, chunk, axes = _lattice_fixture()
    net, codes = U.build_trained_npm(device=dev)
    assert U.state_hash(net) == str(fx["npm_sha256"])
    ck = np.load(os.path.join(U.GOLDEN, "trained_npm_state.npz"))
    code = codes[[int(c) for c in ck["code_ids"]].index(int(fx["npm_code"]))]
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None].to(dev)
    assert net.numerics == "auto"
    vol = R.get_logits(net, code, grid, nbatch_points=chunk)
    rep = net.last_numerics
    e = U.maxdiff(vol, fx["npm_logits"])
    print(f"trained NPM net, {res}^3 get_logits, auto tiers (single-term layers {rep.get('single_mask', 0):#x}, two-term {rep.get('mask', 0):#x}): "
          f"max |hip - reference| = {e:.2e}")
    assert rep.get("single_mask", 0) != 0 or rep.get("mask", 0) != 0
    assert e <= 1e-5


@pytest.mark.gpu
def test_auto_numerics_of_the_identity_field_match_the_reference_at_lattice_size(dev):
    """get_logits (eval mode, chunk 25 000: the overwrite voxels included) and get_logits_backward of the trained pair on the
    64^3 lattice, both with the default numerics - the identity field's calibrated knobs, the deformation stage's tiers"""
    fx, res, chunk, axes = _lattice_fixture()
    inet, codes = U.build_trained_identity(device=dev)
    inet.eval()
    assert U.state_hash(inet) == str(fx["identity_sha256"])
    dnet, z_ex, pairs = U.build_trained_deformation(device=dev)
    i = pairs.index(tuple(int(v) for v in fx["pair"]))
    lat_id = codes[pairs[i][0]]
    grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None].to(dev)
    vol = R.get_logits(inet, lat_id, grid, nbatch_points=chunk)
    c = inet.calibration
    assert inet.numerics == "auto" and c is not None and (c["prune_tol"] > 0 or c["precision"] != "f16x3")     # calibrated knobs ran
    e1 = U.maxdiff(vol, fx["identity_logits"])
    vol2 = R.get_logits_backward(inet, dnet, lat_id, torch.cat([lat_id, z_ex[i]]), grid, nbatch_points=chunk,
                                 anchors=torch.from_numpy(fx["anchors"]).to(dev))
    e2 = U.maxdiff(vol2, fx["two_stage_logits"])
    print(f"trained identity field, {res}^3, auto ({c['precision']}, light {c['light_tol']}, mid {c['mid_tol']}, prune {c['prune_tol']:g}): "
          f"get_logits max |hip - reference| = {e1:.2e}, get_logits_backward {e2:.2e}")
    assert e1 <= 1e-5 and e2 <= 1e-5
