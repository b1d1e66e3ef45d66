"""CPU check of the algebra behind ident_train_kernel.hip: the gradient of
    phi = sbar * f(c) + v . grad_c f(c)
w.r.t. the coordinates and every weight of one member MLP equals the reverse sweep of a forward pass that carries the
value stream and ONE tangent stream along v, in the kernels' scaled domain (layout.h: activations times
k = 100 / ln 2, softplus in base 2).  Compared with torch.autograd's double backward in float64."""
import math

import torch


def test_two_stream_reverse_sweep_equals_double_backward():
    torch.manual_seed(0)
    dt = torch.float64
    H, L1, P = 20, 11, 7
    k, ln2, r2 = 100 / math.log(2), math.log(2), 1 / math.sqrt(2)
    W0c = torch.randn(H, 3, dtype=dt) * 0.5; b0f = torch.randn(H, dtype=dt) * 0.02
    W1 = torch.randn(L1, H, dtype=dt) * 0.3; b1 = torch.randn(L1, dtype=dt) * 0.02
    W2a = torch.randn(H, L1, dtype=dt) * 0.3; W2c = torch.randn(H, 3, dtype=dt) * 0.5; b2f = torch.randn(H, dtype=dt) * 0.02
    W3 = torch.randn(H, H, dtype=dt) * 0.3; b3 = torch.randn(H, dtype=dt) * 0.02
    w4 = torch.randn(H, dtype=dt); b4 = torch.randn((), dtype=dt)
    params = [W0c, b0f, W1, b1, W2a, W2c, b2f, W3, b3, w4, b4]
    for p in params:
        p.requires_grad_()
    c = (torch.randn(P, 3, dtype=dt) * 0.05).requires_grad_()
    sbar = torch.randn(P, dtype=dt)
    v = torch.randn(P, 3, dtype=dt)
    sp = torch.nn.Softplus(beta=100)

    def f(c):       # one member of EnsembledDeepSDF.forward with the latent folded into b0f / b2f
        h0 = sp(c @ W0c.T + b0f)
        h1 = sp(h0 @ W1.T + b1)
        x2 = torch.cat([h1, c], -1) / math.sqrt(2)
        h2 = sp(x2 @ torch.cat([W2a, W2c], 1).T + b2f)
        h3 = sp(h2 @ W3.T + b3)
        return h3 @ w4 + b4

    val = f(c)
    g = torch.autograd.grad(val.sum(), c, create_graph=True)[0]
    phi = (sbar * val).sum() + (v * g).sum()
    ref = torch.autograd.grad(phi, [c] + params)

    with torch.no_grad():
        sp2 = lambda d: torch.clamp(d, min=0) + torch.log2(1 + torch.exp2(-d.abs()))
        sg2 = lambda d: 1 / (1 + torch.exp2(-d))

        def act(d, t):
            s = sg2(d)
            return s, sp2(d), s * t, ln2 * s * (1 - s) * t          # sigma', h', u', sigma'' tau

        s0, h0, u0, q0 = act(k * (c @ W0c.T + b0f), k * (v @ W0c.T))
        s1, h1, u1, q1 = act(h0 @ W1.T + k * b1, u0 @ W1.T)
        s2, h2, u2, q2 = act(r2 * (h1 @ W2a.T) + k * r2 * (c @ W2c.T) + k * b2f, r2 * (u1 @ W2a.T) + k * r2 * (v @ W2c.T))
        s3, h3, u3, q3 = act(h2 @ W3.T + k * b3, u2 @ W3.T)
        assert (h3 @ (w4 / k) + b4 - val).abs().max() < 1e-12
        assert (u3 @ (w4 / k) - (v * g).sum(-1)).abs().max() < 1e-12
        H3, U3 = sbar[:, None] * (w4 / k), (w4 / k).expand(P, H)
        D3, T3 = H3 * s3 + U3 * q3, U3 * s3
        H2, U2 = D3 @ W3, T3 @ W3
        D2, T2 = H2 * s2 + U2 * q2, U2 * s2
        H1, U1, cbar = r2 * (D2 @ W2a), r2 * (T2 @ W2a), k * r2 * (D2 @ W2c)
        D1, T1 = H1 * s1 + U1 * q1, U1 * s1
        H0, U0 = D1 @ W1, T1 @ W1
        D0, T0 = H0 * s0 + U0 * q0, U0 * s0
        cbar = cbar + k * (D0 @ W0c)
        mine = [cbar,
                k * (D0.T @ c + T0.T @ v), k * D0.sum(0),
                D1.T @ h0 + T1.T @ u0, k * D1.sum(0),
                r2 * (D2.T @ h1 + T2.T @ u1), k * r2 * (D2.T @ c + T2.T @ v), k * D2.sum(0),
                D3.T @ h2 + T3.T @ u2, k * D3.sum(0),
                (sbar @ h3 + u3.sum(0)) / k, sbar.sum()]
    for a, b in zip(mine, ref):
        assert (a - b).abs().max() <= 1e-12 * (1 + b.abs().max())


def test_training_work_lists_cover_every_kept_pair_once(monkeypatch):
    """Host-side work lists of the training tier: forward tiles (<= 64 points) and backward tiles (<= 32) over one
    point list ordered by (member, row); pieces and weight-gradient chunks tile the backward table exactly, every
    chunk inside one weight set and one piece."""
    import numpy as np

    import nphm_amd.ensembled_deepsdf as E
    monkeypatch.setattr(E, "_TRAIN_RING_TILES", 7)
    monkeypatch.setattr(E, "_WGRAD_CHUNK", 3)
    torch.manual_seed(0)
    B, N, A = 2, 200, 40
    anch, xyz = torch.randn(B, 39, 3) * 0.1, torch.randn(B, N, 3) * 0.15
    sets = E._member_sets(A, 16)
    _, mask = E._blend_mask(anch, xyz, 1e-7, A)
    tiles_fwd, tiles, plist, chunks, pieces, (edge_tabs, n_sets, ring) = E._train_member_lists(mask, sets)
    tiles_fwd, tiles, plist, chunks, edge_tabs = (t.numpy() for t in (tiles_fwd, tiles, plist, chunks, edge_tabs))
    for tab, width in ((tiles_fwd, 64), (tiles, 32)):
        seen = np.zeros((B, N, A), int)
        assert (tab[:, 3] > 0).all() and (tab[:, 3] <= width).all()
        assert (np.diff(tab[:, 1]) >= 0).all()                              # member-major
        for row, k, off, cnt in tab:
            seen[row, plist[off:off + cnt], k] += 1
        assert (seen == mask.numpy().astype(int)).all()
    T = tiles.shape[0]
    cover = np.zeros(T, int)
    assert sum(n for _, n, _, _ in pieces) == T
    for t0, nt, c0, nc in pieces:
        assert nt <= 7
        for s_, rel, n, _ in chunks[c0:c0 + nc]:
            assert 0 < n <= 3 and rel + n <= nt
            assert (sets.numpy()[tiles[t0 + rel:t0 + rel + n, 1]] == s_).all()
            cover[t0 + rel:t0 + rel + n] += 1
    assert (cover == 1).all()
    # the tables of the edge-gradient kernels: a set's chunks are consecutive in the chunk table, a pair's tiles in the tile table
    assert ring == 7 and n_sets == int(sets.max()) + 1
    set_chunk_first, pair_first = edge_tabs[:n_sets + 1], edge_tabs[n_sets + 1:]
    assert set_chunk_first[0] == 0 and set_chunk_first[-1] == len(chunks) and (np.diff(set_chunk_first) >= 0).all()
    for s_ in range(n_sets):
        assert (chunks[set_chunk_first[s_]:set_chunk_first[s_ + 1], 0] == s_).all()
    first_abs = chunks[:, 3] * ring + chunks[:, 1]
    assert (np.diff(first_abs) > 0).all()                                   # ordered by tile
    assert len(pair_first) == A * B + 1 and pair_first[0] == 0 and pair_first[-1] == T
    for pr in range(A * B):
        sub = tiles[pair_first[pr]:pair_first[pr + 1]]
        assert (sub[:, 1] == pr // B).all() and (sub[:, 0] == pr % B).all()


def test_fused_blend_backward_formulas():
    """nphm_identity_blend_backward (ident_train_kernel.hip): gradients of  L = pbar . pred + qbar . d pred/d x  w.r.t.
    the member values S, their gradients G, the query point and the anchors, with pred the Gaussian blend of
    EnsembledDeepSDF.py:129-150 - closed forms against float64 autograd."""
    torch.manual_seed(0)
    dt = torch.float64
    P, K = 6, 5
    sigma, eps = 0.01, 1e-5
    x = (torch.randn(P, 3, dtype=dt) * 0.1).requires_grad_()
    a = (torch.randn(K, 3, dtype=dt) * 0.1).requires_grad_()
    S = torch.randn(P, K + 1, dtype=dt, requires_grad=True)
    G = torch.randn(P, K + 1, 3, dtype=dt, requires_grad=True)
    pbar, qbar = torch.randn(P, dtype=dt), torch.randn(P, 3, dtype=dt)
    w_bg = math.exp(-0.2 / sigma)
    x0 = x.detach().clone()

    def field(xx):           # blend of member values that have gradient G at x0 (and no curvature)
        d = (a[None] - xx[:, None]).norm(dim=2)
        w = torch.exp(-((d + eps) ** 2) / sigma)
        w = torch.cat([w, torch.full_like(w[:, :1], w_bg)], 1)
        what = w / (w.sum(1, keepdim=True) + 1e-6)
        return (what * (S + (G * (xx - x0)[:, None, :]).sum(-1))).sum(1)

    p = field(x)
    q = torch.autograd.grad(p.sum(), x, create_graph=True)[0]
    gx, ga, gS, gG = torch.autograd.grad((pbar * p).sum() + (qbar * q).sum(), [x, a, S, G])
    with torch.no_grad():
        e = x0[:, None, :] - a[None]
        r = e.norm(dim=2)
        w = torch.exp(-(r + eps) ** 2 / sigma)
        D = w.sum(1) + w_bg + 1e-6
        wh = torch.cat([w, torch.full_like(w[:, :1], w_bg)], 1) / D[:, None]
        c = -2 * (r + eps) / (sigma * r)
        u = torch.cat([c[..., None] * e, torch.zeros(P, 1, 3, dtype=dt)], 1)
        pred = (wh * S).sum(1)
        m, gt = (wh[..., None] * u).sum(1), (wh[..., None] * G).sum(1)
        dS = S - pred[:, None]
        qw = ((dS * wh)[..., None] * u).sum(1)
        assert (pred - p).abs().max() < 1e-12 and (gt + qw - q).abs().max() < 1e-11
        t = (qbar[:, None, :] * u).sum(-1)
        tau, theta = (wh * t).sum(1), (qbar * qw).sum(-1)
        Sbar = wh * (pbar[:, None] + t - tau[:, None])
        Gbar = wh[..., None] * qbar[:, None, :]
        qG, qgt = (qbar[:, None, :] * G).sum(-1), (qbar * gt).sum(-1)
        kappa, qe = 2 * eps / (sigma * r ** 3), (qbar[:, None, :] * e).sum(-1)
        coef = pbar[:, None] * dS[:, :K] + (qG[:, :K] - qgt[:, None]) + dS[:, :K] * (t[:, :K] - tau[:, None]) - theta[:, None]
        dLde = (wh[:, :K] * coef)[..., None] * u[:, :K] + (dS[:, :K] * wh[:, :K])[..., None] * (c[..., None] * qbar[:, None, :] + (kappa * qe)[..., None] * e)
        rel = lambda A, B: float((A - B).abs().max() / B.abs().max())
        assert rel(Sbar, gS) < 1e-12 and rel(Gbar, gG) < 1e-12 and rel(-dLde.sum(0), ga) < 1e-12
        # autograd's d/dx also holds the member path sum_k Sbar_k G_k (the member kernels' share)
        assert rel(dLde.sum(1) + (Sbar[..., None] * G).sum(1), gx) < 1e-12
