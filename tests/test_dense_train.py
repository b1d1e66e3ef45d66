"""GPU parity of the dense training tier (widened in round 6): the DeepSDF backbone of the forward-deformation
network with TRAINABLE parameters on csrc/dense_train_kernels.hip - what scripts/training/train_corresp.py trains through
compute_loss_corresp_forward (src/NPHM/models/loss_functions.py:282-326).  Reference = the composite tier (the reference's
nn.Linear / Softplus sequence under PyTorch autograd) on the same inputs and, for the loss, the fixture the REFERENCE's own
function wrote (tests/golden/training_def.npz).  Tolerances are relative to a tensor's largest entry."""
import numpy as np
import pytest
import torch

import _util as U
from nphm_amd import _lib
from nphm_amd.deepsdf import _DenseLayerFn
from nphm_amd.loss_functions import compute_loss_corresp_forward

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("M,N,K,epi,e_rows", [(1000, 512, 512, 1, 1000), (777, 277, 512, 1, 259), (130, 512, 235, 1, 130),
                                               (515, 300, 3, 2, 1), (64, 3, 512, 0, 1), (2050, 512, 280, 1, 1)])
def test_gemm_nt_with_epilogues_against_float64(dev, M, N, K, epi, e_rows):
    """C = act(alpha A B^T + E[m / e_rows]) for ragged shapes (rows / columns / K that fill no tile, leading dimensions that
    rule out 16-byte loads): split-bf16 x3 (16 product bits, fp32 accumulation) against a float64 product - 5.5e-6 of the
    largest entry measured at K = 512, asserted 1.5e-5"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) * 0.7).to(dev)
    B = (torch.randn(N, K, generator=g) * 0.3).to(dev)
    n_e = (M + e_rows - 1) // e_rows
    E = torch.randn(n_e, N, generator=g).to(dev)
    alpha, beta = 0.70710678, 100.0
    C = torch.full((M, N), float("nan"), device=dev)
    _lib.check(lib.nphm_dense_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, E.data_ptr() if epi else None, e_rows,
                                      alpha, beta, epi, 1, None), "gemm")
    ref = alpha * (A.double() @ B.double().t())
    if epi:
        ref = ref + E.double().repeat_interleave(e_rows, dim=0)[:M]
    if epi == 1:
        ref = torch.nn.functional.softplus(ref, beta=beta)
    assert bool(torch.isfinite(C).all())
    assert _rel(C, ref) < 1.5e-5
    # the split form: the K range cut into pieces, added in order
    if epi == 0 or K >= 64:
        parts = torch.empty(4, M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        _lib.check(lib.nphm_dense_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, parts.data_ptr(), N, M, N, K, None, 1, 1.0, 0.0, 0, 4, None), "gemm")
        _lib.check(lib.nphm_dense_reduce_splits(parts.data_ptr(), 4, M * N, alpha, out.data_ptr(), None), "reduce")
        assert _rel(out, alpha * (A.double() @ B.double().t())) < 1.5e-5


@pytest.mark.parametrize("M,K,N,beta", [(1000, 512, 512, 100.0), (333, 280, 277, 100.0), (70, 235, 512, None), (515, 64, 40, 0.0)])
def test_dense_layer_forward_and_gradients_against_autograd(dev, M, K, N, beta):
    g = torch.Generator().manual_seed(7 * M + K)
    x = (torch.randn(M, K, generator=g) * 0.5).to(dev).requires_grad_()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).requires_grad_()
    b = (torch.randn(N, generator=g) * 0.1).to(dev).requires_grad_()
    go = torch.randn(M, N, generator=g).to(dev)
    alpha = 0.70710678
    y = _DenseLayerFn.apply(x, W, b, alpha, beta)
    y.backward(go)
    got = [y.detach(), x.grad.clone(), W.grad.clone(), b.grad.clone()]
    xd, Wd, bd = [t.detach().double().requires_grad_() for t in (x, W, b)]
    pre = alpha * xd @ Wd.t() + bd
    ref = pre if beta is None else (torch.nn.functional.softplus(pre, beta=beta) if beta > 0 else torch.relu(pre))
    ref.backward(go.double())
    want = [ref.detach(), xd.grad, Wd.grad, bd.grad]
    errs = [_rel(a, b_) for a, b_ in zip(got, want)]
    print(f"dense layer {M}x{K}->{N}, beta {beta}: y {errs[0]:.1e}, dx {errs[1]:.1e}, dW {errs[2]:.1e}, db {errs[3]:.1e}")
    assert max(errs) < 3e-5
    # fixed-order sums: a second backward gives the same bits
    x.grad = W.grad = b.grad = None
    _DenseLayerFn.apply(x, W, b, alpha, beta).backward(go)
    assert torch.equal(W.grad, got[2]) and torch.equal(b.grad, got[3]) and torch.equal(x.grad, got[1])


def _corresp_step(dev, backend, seed=11):
    shape_net = U.build_identity(device=dev).train()
    expr_net = U.build_deformation(device=dev).train()
    expr_net.defDeepSDF.train_backend = backend
    g = torch.Generator().manual_seed(5)
    B, n = 4, 500
    lat_shape, lat_expr = torch.nn.Embedding(6, 1344).to(dev), torch.nn.Embedding(9, 200).to(dev)
    with torch.no_grad():
        lat_shape.weight.copy_(torch.stack([U.sample_latent(30 + i) for i in range(6)]).to(dev))
        lat_expr.weight.copy_((torch.randn(9, 200, generator=g) * 0.1).to(dev))
    neutral = ((torch.rand(B, n, 3, generator=g) - 0.5) * torch.tensor([0.5, 0.6, 0.5]))
    batch = {"points_neutral": neutral, "points_posed": torch.cat([neutral + 0.01 * torch.randn(B, n, 3, generator=g),
                                                                    torch.randn(B, n, 3, generator=g)], -1),
             "gt_anchors": torch.from_numpy(U.anchors_mean()).reshape(1, 39, 3).repeat(B, 1, 1),
             "subj_ind": torch.tensor([[2], [0], [5], [3]]), "idx": torch.tensor([[8], [1], [2], [4]])}
    used = []
    orig = expr_net.defDeepSDF.evaluate_train_hip
    expr_net.defDeepSDF.evaluate_train_hip = lambda *a, **k: used.append(1) or orig(*a, **k)
    torch.manual_seed(seed)                      # the conditioning noise and the free samples: same device stream for both tiers
    losses = compute_loss_corresp_forward(batch, expr_net, shape_net, lat_expr, lat_shape, dev, epoch=3)
    (100.0 * losses["corresp"] + 0.01 * losses["lat_reg"] + 5.0 * losses["loss_reg_zero"]).backward()
    out = {"loss." + k: v.detach() for k, v in losses.items()}
    out["expr_table"] = lat_expr.weight.grad.clone()
    out["shape_table"] = lat_shape.weight.grad.clone()
    for name, p in list(expr_net.named_parameters()) + [("mlp_pos." + n_, q) for n_, q in shape_net.mlp_pos.named_parameters()]:
        out[name] = torch.zeros_like(p) if p.grad is None else p.grad.clone()
    return out, len(used)


def test_train_corresp_step_against_the_composite_tier(dev):
    """One step of the second training stage (nphm_def.yaml lambdas): losses, every parameter gradient of the deformation
    network (backbone AND compressor), the identity decoder's anchor head and both code tables - dense training tier against
    the composite tier (PyTorch autograd over the reference's op sequence), <= 2e-4 of each tensor's largest entry."""
    ref, used_ref = _corresp_step(dev, "composite")
    got, used = _corresp_step(dev, "hip")
    assert used_ref == 0 and used == 2, (used_ref, used)          # the tier ran: neutral points + free samples
    worst = {k: _rel(got[k], ref[k]) for k in ref}
    print("train_corresp step, dense training tier against composite:", {k: f"{v:.1e}" for k, v in worst.items() if v > 1e-6})
    bad = {k: v for k, v in worst.items() if not v < 2e-4}
    assert not bad, bad
    assert all(float(v.abs().max()) > 0 for k, v in got.items() if k.startswith("defDeepSDF.lin"))


def test_compute_loss_corresp_forward_matches_the_reference_fixture_on_the_gpu(dev, monkeypatch):
    """The REFERENCE's compute_loss_corresp_forward wrote tests/golden/training_def.npz on the CPU; here the same call runs on
    the GPU with the dense training tier.  Its random draws (conditioning noise, free samples) are taken from the CPU generator -
    the stream the reference consumed - and moved to the device, so losses and gradients are comparable number by number."""
    from test_training_loss import _batch, _grad_norms, _tables
    g = U.golden("training_def")
    rn, rr = torch.randn, torch.rand
    monkeypatch.setattr(torch, "randn", lambda *s, device=None, **k: rn(*s, **k).to(device) if device is not None else rn(*s, **k))
    monkeypatch.setattr(torch, "rand", lambda *s, device=None, **k: rr(*s, **k).to(device) if device is not None else rr(*s, **k))
    shape_net = U.build_identity(device=dev).train()
    shape_net.prune_tol = -1.0
    expr_net = U.build_deformation(device=dev).train()
    assert U.state_hash(expr_net) == str(g["c_state_hash_expr"])
    lat_shape, lat_expr = _tables(g, dev)
    used = []
    orig = expr_net.defDeepSDF.evaluate_train_hip
    expr_net.defDeepSDF.evaluate_train_hip = lambda *a, **k: used.append(1) or orig(*a, **k)
    torch.manual_seed(int(g["seed_call"]))
    losses = compute_loss_corresp_forward(_batch(g, "c_batch_", dev), expr_net, shape_net, lat_expr, lat_shape, dev, epoch=3)
    sum(losses.values()).backward()
    assert len(used) == 2
    for k, v in losses.items():
        assert abs(float(v.detach()) - float(g["c_loss_" + k])) <= 2e-5 * max(1.0, abs(float(g["c_loss_" + k]))), k
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    e_tab = rel(lat_expr.weight.grad.cpu().numpy(), g["c_grad_expr_table"])
    norms = _grad_norms([("expr", expr_net), ("shape", shape_net)])
    e_norm = float(np.abs(norms - g["c_grad_norms"]).max() / g["c_grad_norms"].max())
    print(f"compute_loss_corresp_forward on the GPU against the reference fixture: expression-code gradients {e_tab:.1e}, "
          f"parameter gradient norms {e_norm:.1e}")
    assert e_tab < 2e-4 and e_norm < 2e-4


def test_gemm_function_is_differentiable_twice(dev):
    """_GemmNTFn's backward is made of _GemmNTFn products: a loss on the gradient w.r.t. the input (the eikonal pattern of
    loss_joint / the NPM trainer: gradient(sdf, x, create_graph=True), then backward()) against float64 autograd."""
    from nphm_amd.deepsdf import _GemmNTFn
    g = torch.Generator().manual_seed(3)
    M, K, N = 4100, 259, 400
    x0 = (torch.randn(M, K, generator=g) * 0.3).to(dev)
    W0 = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    V0 = (torch.randn(3, N, generator=g) / N ** 0.5).to(dev)

    def run(x, W, V, mm):
        h = torch.nn.functional.softplus(mm(x, W), beta=100.0)
        out = h @ V.t()
        (gx,) = torch.autograd.grad(out.sum(), x, create_graph=True)
        loss = (gx.norm(dim=-1) - 1).abs().mean() + out.square().mean()
        loss.backward()
        return loss.detach(), x.grad, W.grad, V.grad

    a = [t.clone().requires_grad_() for t in (x0, W0, V0)]
    got = run(*a, lambda x, W: _GemmNTFn.apply(x, W, 1.0))
    b = [t.double().clone().requires_grad_() for t in (x0, W0, V0)]
    want = run(*b, lambda x, W: x @ W.t())
    errs = [_rel(p, q) for p, q in zip(got, want)]
    print("second-order pass through _GemmNTFn against float64: loss %.1e, dx %.1e, dW %.1e, dV %.1e" % tuple(errs))
    assert max(errs) < 3e-4          # (dx carries sigma'' = 25 at beta 100 on top of the products' 16 bits: 1.1e-4 measured)


def test_second_order_pass_through_the_backbone_uses_the_kernels(dev):
    """A DeepSDF whose query points need a gradient that is differentiated again (first-order tier refuses): `evaluate` with
    its hidden products on _GemmNTFn, against the composite tier - values, d/dx and every parameter gradient of an eikonal-style
    loss."""
    import nphm_amd
    torch.manual_seed(0)
    net = nphm_amd.DeepSDF(lat_dim=64, hidden_dim=400, nlayers=4, out_dim=3).to(dev).train()
    g = torch.Generator().manual_seed(9)
    xyz = ((torch.rand(3, 900, 3, generator=g) - 0.5)).to(dev)
    lat = (torch.randn(3, 1, 64, generator=g) * 0.3).to(dev)

    def run(backend):
        net.train_backend = backend
        net.zero_grad(set_to_none=True)
        x = xyz.clone().requires_grad_()
        l = lat.clone().requires_grad_()
        calls = []
        orig = nphm_amd.deepsdf._GemmNTFn.apply
        out, _ = net(x, l.expand(3, 900, 64))
        (gx,) = torch.autograd.grad(out[..., 0].sum(), x, create_graph=True)
        ((gx.norm(dim=-1) - 1).abs().mean() + out.square().mean()).backward()
        return [out.detach(), gx.detach(), l.grad] + [p.grad.clone() for p in net.parameters()]

    ref = run("composite")
    got = run("hip")
    errs = [_rel(a, b) for a, b in zip(got, ref)]
    print("second-order pass through DeepSDF, kernels against composite:", ["%.1e" % e for e in errs])
    assert max(errs) < 2e-4 and max(errs) > 0          # (> 0: the kernel path did run - its products carry 16 bits)
