"""Latent-code fitting loops — host-side mirror of src/NPHM/models/fitting.py
(``inference_iterative_root_finding_joint`` :14-177, ``inference_identity_space`` :180-288):
Adam on the identity code (and one expression code per observation) so that the observed points lie
on the zero level set of the posed SDF.  The loops are CALLERS of the hot path:
per step they run one anchor forward, a Broyden correspondence search through the deformation field
(no-grad forwards -> fused HIP kernel), two Jacobians of the deformation field and one
forward/backward of the identity field (autograd -> composite tier).

Same signatures, RNG consumption order, schedules and loss terms as the reference; additions are
keyword-only (``verbose``, ``history``)."""
from __future__ import annotations

from contextlib import contextmanager, nullcontext
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import optim

from .diff_operators import inverse3x3 as _inverse3x3
from .iterative_root_finding import jac, nabla, search

_UNOBSERVED = (30, 31, 39)          # local codes that the single-view scans never see (fitting.py:148)


@contextmanager
def _frozen(*modules):
    """The fitting loops optimise latent codes only; the reference leaves the decoders' parameters
    trainable and lets autograd fill their .grad, which nothing reads.  Freezing them
    for the duration of the loop skips that work and lets the identity field use its hand-written
    first-order backward; the flags are restored afterwards.  The fitted latents do not change."""
    saved = [(p, p.requires_grad) for m in modules if m is not None for p in m.parameters()]
    for p, _ in saved:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, flag in saved:
            p.requires_grad_(flag)


def _anchors_of(decoder, lat_rep_shape, device):
    """``_, anchors = decoder(zeros[1,1,3], lat, None)`` (fitting.py:58, :208): the mirrored identity net
    answers from its anchor head alone; any other decoder gets the reference's one-point forward."""
    if hasattr(decoder, "predict_anchors"):
        return decoder.predict_anchors(lat_rep_shape)
    return decoder(torch.zeros([1, 1, 3], device=device), lat_rep_shape, None)[1]


def _apply_schedule(j, step_scale, schedule_cfg, lambdas, optimizers, with_expr):
    """Hand-tuned learning-rate / loss-weight schedule (fitting.py:40-52, :197-206)."""
    key = int(j / step_scale)
    if key in schedule_cfg["lr"]:
        for o in optimizers:
            for group in o.param_groups:
                group["lr"] /= schedule_cfg["lr"][key]
    if key in schedule_cfg["symm_dist"]:
        lambdas["symm_dist"] /= schedule_cfg["symm_dist"][key]
    if key in schedule_cfg["reg_glob"]:
        lambdas["reg_global"] /= schedule_cfg["reg_glob"][key]
    if key in schedule_cfg["reg_loc"]:
        lambdas["reg_loc"] /= schedule_cfg["reg_loc"][key]
    if with_expr and key in schedule_cfg["reg_expr"]:
        lambdas["reg_expr"] /= schedule_cfg["reg_expr"][key]


class _CodeAdam(optim.Adam):
    """optim.Adam(params=[code], lr=lr) as the reference builds it (fitting.py:47-48, :196) - same defaults, same
    ``param_groups`` (the schedules write ``lr`` there) and ``state`` entries - whose ``step`` on a ROCm fp32 code tensor is
    ONE launch (``nphm_adam_step``) instead of the multi-tensor implementation's eight: behind a replayed hipGraph the
    step is bound by the GPU's launch rate, and the two optimizers' sixteen launches were 3 % of it.  Same update rule
    (exp_avg.lerp_, exp_avg_sq.mul_.addcmul_, addcdiv_ with host-side bias corrections); other tensors: the parent's step."""

    def _fast_ok(self):
        return all(
            p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and (p.grad is None or (p.grad.is_contiguous() and not p.grad.is_sparse))
            for g in self.param_groups for p in g["params"]) and all(
            not g["amsgrad"] and g["weight_decay"] == 0 and not g["maximize"] and not g.get("capturable") and not g.get("differentiable")
            for g in self.param_groups)

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)          # a host tensor, as the parent keeps it
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        if not (closure is None and self._fast_ok()):
            return super().step(closure)
        import math
        from . import _lib
        lib = _lib.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st["step"] += 1
                t = float(st["step"])
                lr = float(group["lr"])
                _lib.check(lib.nphm_adam_step(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                              p.numel(), b1, b2, lr / (1.0 - b1 ** t), math.sqrt(1.0 - b2 ** t), group["eps"],
                                              torch.cuda.current_stream(p.device).cuda_stream), "nphm_adam_step")
                # the kernel wrote through the raw pointer: tell autograd (and every cache keyed on the code's version
                # counter - anchor_scope, DeformationNetwork.prime_condition, prepare_latent's state scope) that it changed
                torch.autograd.graph.increment_version(p)
        return None


def _adam(code, lr):
    return _CodeAdam(params=[code], lr=lr) if code.is_cuda else optim.Adam(params=[code], lr=lr)


class _PairAdam:
    """``opt.step(); opt_expr.step()`` of the joint loop (fitting.py:170-171) as ONE launch INSIDE the step - and so inside its
    replayed hipGraph (``nphm_adam_step_pair``).  What changes from step to step - the bias corrections and the scheduled
    learning rate - is computed on the host as before (``host_scalars``: advances the optimizers' step counts, same float32
    values ``nphm_adam_step`` receives) and travels to the device with the step's draw, in the same pinned upload; the kernel
    reads it from there.  Same state entries, same update rule, same bits as two ``_CodeAdam.step()`` calls."""

    SLOTS = 6                        # int64 slots behind the draw = 2 x 6 float32

    def __init__(self, optimizers):
        import os
        self.opts = list(optimizers)                 # (one or two: the identity-only loop has one code)
        self.ok = (len(self.opts) in (1, 2) and os.environ.get("NPHM_AMD_FIT_FUSED", "1") not in ("0", "")
                   and all(isinstance(o, _CodeAdam) and o._fast_ok() and len(o.param_groups) == 1 and len(o.param_groups[0]["params"]) == 1
                           for o in self.opts))

    def params(self):
        return [o.param_groups[0]["params"][0] for o in self.opts]

    @torch.no_grad()
    def host_scalars(self):
        import math
        import numpy as np
        vals = []
        for o in self.opts:
            group = o.param_groups[0]
            st = o._state_of(group["params"][0])
            st["step"] += 1
            t = float(st["step"])
            b1, b2 = group["betas"]
            f = np.float32
            vals += [f(1) - f(b1), f(b2), f(1) - f(b2), f(float(group["lr"]) / (1.0 - b1 ** t)), f(math.sqrt(1.0 - b2 ** t)), f(group["eps"])]
        vals += [np.float32(0)] * (12 - len(vals))
        return torch.from_numpy(np.asarray(vals, dtype=np.float32).view(np.int64).copy())

    @torch.no_grad()
    def launch(self, scalars_i64):
        """the update itself (inside the step, behind loss.backward()); ``scalars_i64``: the tail of the uploaded draw"""
        import ctypes
        from . import _lib
        lib = _lib.load()
        ps = self.params()
        sts = [o.state[p] for o, p in zip(self.opts, ps)]
        pad = lambda ts: list(ts) + [None] * (2 - len(ts))
        arr = lambda ts: (ctypes.c_void_p * 2)(*[None if t is None else t.data_ptr() for t in pad(ts)])
        n = (ctypes.c_int64 * 2)(*[0 if (p is None or p.grad is None) else p.numel() for p in pad(ps)])
        _lib.check(lib.nphm_adam_step_pair(arr(ps), arr([p.grad for p in ps]), arr([st["exp_avg"] for st in sts]),
                                           arr([st["exp_avg_sq"] for st in sts]), n, scalars_i64.data_ptr(),
                                           torch.cuda.current_stream(ps[0].device).cuda_stream), "nphm_adam_step_pair")

    def bump(self):
        """after a step (replayed or eager): the kernel wrote through raw pointers - tell autograd and every cache keyed on the
        codes' version counters"""
        for p in self.params():
            torch.autograd.graph.increment_version(p)


class _ObservationSampler:
    """n_batch observations with replacement, <= n_points points each with replacement (fitting.py:61-70):
    the same torch.randint calls in the same order as the reference (host RNG), but the picked indices travel
    to the device in ONE pinned upload and the points are gathered from one padded [n_obs, P, 3] tensor in one
    indexing kernel (the reference indexes every cloud with a host index tensor: one blocking copy each)."""

    def __init__(self, all_obs, n_batch, n_points):
        self.sizes = [int(c.shape[0]) for c in all_obs]
        self.n_batch, self.n_points = n_batch, n_points
        self.device = all_obs[0].device
        pmax = max(self.sizes)
        self.clouds = torch.zeros(len(all_obs), pmax, all_obs[0].shape[1], dtype=all_obs[0].dtype, device=self.device)
        for i, c in enumerate(all_obs):
            self.clouds[i, : c.shape[0]] = c
        self.on_gpu = self.device.type == "cuda"
        self.extra = 0           # int64 slots behind the indices of an uploaded draw (_PairAdam's scalars)
        self.ring = None         # replayed steps: the draws travel through a ring in pinned host memory (enable_ring)
        self.ring_recorded = self.rows_logged = False
        self.last_seq = None

    RING_SLOTS = 8

    def enable_ring(self, like, log_rows=0):
        """A replayed step reads its draw straight from pinned host memory (``nphm_fit_inputs_ring``) instead of from a device
        buffer that a stream-ordered upload filled in front of the replay: ``RING_SLOTS`` draws of the shape of ``like``, a
        device counter of the replays done (which slot the next replay reads), one event per slot (the host fills slots
        ahead of the device, never the one a replay in flight may still read)."""
        import os
        if not self.on_gpu or os.environ.get("NPHM_AMD_FIT_RING", "1") in ("0", ""):
            return
        try:
            ring = torch.empty(self.RING_SLOTS, like.numel(), dtype=torch.int64).pin_memory()
        except RuntimeError:                                  # no page-locked memory to be had: the upload path stays
            return
        if not ring.is_pinned():
            return
        self.ring = ring
        self.ring_ctl = torch.zeros(2, dtype=torch.int32, device=self.device)
        self.ring_seq = 0                                  # replays issued so far = the value of ring_ctl[0] once they have run
        self.ring_recorded = False                         # the current recording of the step reads the ring
        # the loss rows of the replays, by replay number (nphm_fit_loss_with_gradients_logged), when a trace is kept
        self.row_log = torch.zeros(log_rows, 8, dtype=torch.float32, device=self.device) if log_rows > 0 else None
        self.rows_logged = False                           # the current recording stores them
        self.last_seq = None                               # replay number of the step just run (None: it ran eagerly)
        self.ring_events = [None] * self.RING_SLOTS

    def ring_write(self, drawn):
        slot = self.ring_seq % self.RING_SLOTS
        ev = self.ring_events[slot]
        if ev is not None:
            ev.synchronize()                               # the replay that read this slot RING_SLOTS steps ago has finished
        self.ring[slot].copy_(drawn)

    def ring_commit(self):
        """behind a replay that consumed the slot written last"""
        slot = self.ring_seq % self.RING_SLOTS
        if self.ring_events[slot] is None:
            self.ring_events[slot] = torch.cuda.Event()
        self.ring_events[slot].record()
        self.last_seq = self.ring_seq
        self.ring_seq += 1

    def draw(self):
        """host side: (obs_idx [n_batch], point indices [n_batch, n]) as ONE flat int64 tensor [n_batch | n_batch * n]
        (both parts contiguous on the device: no strided-column copies inside the step)"""
        obs_idx = torch.randint(0, len(self.sizes), [self.n_batch])
        rows = []
        for i in range(self.n_batch):
            size = self.sizes[int(obs_idx[i])]
            rows.append(torch.randint(0, size, [min(self.n_points, size)]))
        if len({r.shape[0] for r in rows}) != 1:
            raise RuntimeError("stack expects each tensor to be equal size (observations of different sizes below "
                               "n_points, as in the reference)")
        return torch.cat([obs_idx, torch.stack(rows, 0).reshape(-1)])

    def draw_like(self):
        """an index tensor of the shape ``draw`` returns, WITHOUT touching the RNG (allocation of the static input)"""
        return torch.zeros(self.n_batch * (1 + min(self.n_points, min(self.sizes))) + self.extra, dtype=torch.int64)

    def upload(self, drawn, out=None):
        if self.on_gpu:
            drawn = drawn.pin_memory()
        if out is None:
            return drawn.to(self.device, non_blocking=True)
        out.copy_(drawn, non_blocking=True)
        return out

    def gather(self, drawn_dev):
        """device side: (obs_idx [n_batch] long, points [n_batch, n, 3])"""
        obs_idx = drawn_dev[: self.n_batch]
        return obs_idx, self.clouds[obs_idx[:, None], drawn_dev[self.n_batch: drawn_dev.numel() - self.extra].view(self.n_batch, -1)]


def _shape_regularisers(decoder, lat_rep_shape, loss_dict):
    """Identity-code regularisers (fitting.py:139-166)."""
    if hasattr(decoder, "lat_dim_glob"):
        loss_dict["reg_loc"] = (torch.norm(lat_rep_shape[..., 64:], dim=-1) ** 2).mean()
        loss_dict["reg_global"] = (torch.norm(lat_rep_shape[..., :64], dim=-1) ** 2).mean()
        loss_dict["reg_unobserved"] = 0
        for idx in _UNOBSERVED:
            loss_dict["reg_unobserved"] += torch.norm(
                lat_rep_shape[..., 64 + idx * 32:64 + (idx + 1) * 32], dim=-1).square().mean()
        g, s, d = decoder.lat_dim_glob, decoder.num_symm_pairs, decoder.lat_dim_loc
        pairs = lat_rep_shape[:, :, g:g + 2 * s * d].view(lat_rep_shape.shape[0], 2 * s, d)
        loss_dict["symm_dist"] = torch.norm(pairs[:, ::2, :] - pairs[:, 1::2, :], dim=-1).mean()
    else:
        loss_dict["symm_dist"] = 0
        loss_dict["reg_unobserved"] = 0
        loss_dict["reg_loc"] = 0
        loss_dict["reg_global"] = (torch.norm(lat_rep_shape, dim=-1) ** 2).mean()


def _report(j, lambdas, loss_dict, extra=None):
    line = "Epoch: {:5d}".format(j)
    for k in lambdas.keys():
        line += " " + k + " {:02.8f} ".format(float(torch.as_tensor(loss_dict[k]).detach()))
    print(line) if extra is None else print(line, extra)


class _History:
    """Per-step loss terms kept on the device and fetched ONCE after the loop (a ``float()`` per term and step
    is a device->host synchronisation each)."""

    def __init__(self, history, keys, n_iter, device, extra=()):
        self.history, self.keys, self.extra = history, list(keys), list(extra)
        self.buf = None if history is None else torch.zeros(max(n_iter, 1), len(self.keys) + 1 + len(self.extra),
                                                            dtype=torch.float32, device=device)

    def row(self, loss_dict, loss, **extra):
        """the step's record as ONE tensor (built inside the step so that it can live in a captured graph)"""
        dev = loss.device
        vals = [torch.as_tensor(loss_dict[k], dtype=torch.float32, device=dev).detach().reshape(()) for k in self.keys]
        vals.append(loss.detach().reshape(()).float())
        vals += [torch.as_tensor(extra[k], device=dev).detach().reshape(()).float() for k in self.extra]
        return torch.stack(vals)

    perm = None          # fused steps hand over the loss kernel's raw row [8]; this is its order in the history's columns
    log = None           # rows of replayed steps, by replay number (written by the step's loss launch: _ObservationSampler.row_log)

    def note(self, j, seq):
        """step j was replay number ``seq``: its row is in ``log`` (no copy launch behind the replay)"""
        if self.buf is not None:
            self._noted = getattr(self, "_noted", [])
            self._noted.append((j, seq))

    def record(self, j, row):
        if self.buf is not None:
            if self.perm is not None and self.buf.shape[1] != row.shape[0]:
                self.buf = torch.zeros(self.buf.shape[0], row.shape[0], dtype=torch.float32, device=self.buf.device)
            self.buf[j].copy_(row)

    def ordered(self, row_host):
        """a step's row on the host in (lambdas order, total, extras)"""
        return row_host if self.perm is None else row_host[self.perm]

    def flush(self, n_done):
        if self.buf is None:
            return
        rows = self.buf[:n_done].cpu().numpy()
        noted = [(j, q) for j, q in getattr(self, "_noted", []) if j < n_done]
        if noted:
            log = self.log.cpu().numpy()
            if rows.shape[1] != log.shape[1]:               # (every step was replayed: record() never widened the buffer)
                rows = np.zeros((rows.shape[0], log.shape[1]), dtype=np.float32)
            for j, q in noted:
                rows[j] = log[q]
        if self.perm is not None:
            rows = rows[:, self.perm]
        for r in rows:
            d = {k: float(v) for k, v in zip(self.keys + ["loss"], r)}
            for k, v in zip(self.extra, r[len(self.keys) + 1:]):
                d[k] = int(round(float(v)))
            self.history.append(d)


class _StepControls:
    """Everything the schedule changes between steps, as device scalars the step READS (so that one captured
    graph serves all steps): the loss weights in ``lambdas`` order and the clamp of the surface loss
    (fitting.py:119-132: |sdf| < 0.1, then < 0.05 after step 250, < 0.0075 after step 500 - nested masks = the
    smallest threshold).  ``refresh`` uploads only when a value changed."""

    def __init__(self, lambdas, device):
        self.keys = list(lambdas.keys())
        self.lam = torch.zeros(len(self.keys), dtype=torch.float32, device=device)
        self.thr = torch.zeros((), dtype=torch.float32, device=device)
        self.one = torch.ones((), dtype=torch.float32, device=device)          # seed of loss.backward
        self.lam6 = torch.zeros(6, dtype=torch.float32, device=device)       # the weights in the fused kernel's term order
        self._lam_host, self._thr_host = None, None

    def refresh(self, lambdas, j, step_scale):
        lam = [float(lambdas[k]) for k in self.keys]
        thr = 0.0075 if j > int(500 * step_scale) else (0.05 if j > int(250 * step_scale) else 0.1)
        if lam != self._lam_host:
            self.lam.copy_(torch.tensor(lam, dtype=torch.float32))
            l6 = [0.0] * 6
            for k, v in zip(self.keys, lam):
                if k in _LOSS_SLOTS:
                    l6[_LOSS_SLOTS[k]] = v
            self.lam6.copy_(torch.tensor(l6, dtype=torch.float32))
            self._lam_host = lam
        if thr != self._thr_host:
            self.thr.fill_(thr)
            self._thr_host = thr

    def fused_perm(self, extra=()):
        """columns of the fused kernel's row [8] in the history's order (lambdas order, total, extras): applied on the host when
        the trace is read - the step itself hands the raw row over (no index_select launch per step)"""
        return [_LOSS_SLOTS[k] for k in self.keys] + [6] + [7] * len(extra)

    def total(self, loss_dict):
        loss = 0
        for i, k in enumerate(self.keys):
            loss = loss + loss_dict[k] * self.lam[i]
        return loss


_LOSS_SLOTS = {"surface": 0, "reg_expr": 1, "reg_global": 2, "reg_unobserved": 3, "reg_loc": 4, "symm_dist": 5}


class _FitLossFn(torch.autograd.Function):
    """Every loss term of a fitting step and their weighted total in ONE launch (``nphm_fit_loss``), the gradients
    w.r.t. the SDF values, the identity code and the expression codes in a second one (fitting.py:115-166; ~100
    elementwise / reduction launches of the PyTorch formulation).  Returns (total, row [8] = the six terms in
    ``_LOSS_SLOTS`` order, the total, the number of valid correspondences)."""

    @staticmethod
    def forward(ctx, sdf, valid, z_shape, z_expr, obs_idx, thr, lam6, seed=None, log=None):
        """``seed``: the device scalar the caller will pass to ``loss.backward(gradient=seed)`` right behind this call - the
        gradients are then computed by THIS launch (nphm_fit_loss_with_gradients) and ``backward`` hands them over when it
        receives that very tensor (one launch per step instead of two; anything else: the backward kernel as before)."""
        from . import _lib
        lib = _lib.load()
        dev = sdf.device
        sdf_c = sdf.detach().reshape(-1).contiguous()
        valid_c = None if valid is None else valid.reshape(-1).contiguous()
        zs = z_shape.detach().reshape(-1).contiguous()
        ze = None if z_expr is None else z_expr.detach().contiguous()
        obs_idx = None if obs_idx is None else obs_idx.to(torch.int64).contiguous()       # a strided column of the draw
        buf = torch.empty(9, dtype=torch.float32, device=dev)           # the report row [8] | the total once more
        row = buf[:8]
        stream = torch.cuda.current_stream(dev).cuda_stream
        n_obs, expr_dim = (ze.shape[0], ze.shape[-1]) if ze is not None else (0, 0)
        ctx.seeded = None
        if seed is not None and seed.is_cuda and seed.dtype == torch.float32 and seed.numel() == 1:
            g_sdf, g_shape = torch.empty_like(sdf_c), torch.empty_like(zs)
            g_expr = None if ze is None else torch.empty_like(ze)
            # ``log`` (the sampler of a loop whose replays read the host ring): while the step is RECORDED the row is also
            # stored in its replay-indexed log (no copy launch between two replays; _History reads it after the loop)
            logged = log is not None and getattr(log, "row_log", None) is not None and torch.cuda.is_current_stream_capturing()
            _lib.check(lib.nphm_fit_loss_with_gradients_logged(
                sdf_c.data_ptr(), None if valid_c is None else valid_c.data_ptr(), sdf_c.numel(), thr.data_ptr(), lam6.data_ptr(),
                zs.data_ptr(), None if ze is None else ze.data_ptr(), None if ze is None else obs_idx.data_ptr(),
                0 if ze is None else obs_idx.numel(), n_obs, expr_dim, seed.data_ptr(), row.data_ptr(), g_sdf.data_ptr(),
                g_shape.data_ptr(), None if g_expr is None else g_expr.data_ptr(),
                log.row_log.data_ptr() if logged else None, log.ring_ctl.data_ptr() if logged else None,
                log.row_log.shape[0] if logged else 0, stream), "nphm_fit_loss_with_gradients")
            if logged:
                log.rows_logged = True
            ctx.seeded = (seed.data_ptr(), seed._version, g_sdf, g_shape, g_expr)
        else:
            _lib.check(lib.nphm_fit_loss(sdf_c.data_ptr(), None if valid_c is None else valid_c.data_ptr(), sdf_c.numel(), thr.data_ptr(),
                                         lam6.data_ptr(), zs.data_ptr(), None if ze is None else ze.data_ptr(),
                                         None if ze is None else obs_idx.data_ptr(), 0 if ze is None else obs_idx.numel(), n_obs, expr_dim,
                                         row.data_ptr(), stream), "nphm_fit_loss")
        ctx.save_for_backward(sdf_c, valid_c, zs, ze, obs_idx, thr, lam6)
        ctx.shapes = (sdf.shape, z_shape.shape, None if z_expr is None else z_expr.shape)
        ctx.mark_non_differentiable(row)
        ctx.set_materialize_grads(False)       # (no zeros [8] + fill launch for the report row's absent gradient)
        return buf[8], row

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_total, _g_row):
        if g_total is None:
            return None, None, None, None, None, None, None, None, None
        s_sdf, s_shape, s_expr = ctx.shapes
        if ctx.seeded is not None and g_total.data_ptr() == ctx.seeded[0] and g_total._version == ctx.seeded[1]:
            _, _, g_sdf, g_shape, g_expr = ctx.seeded          # the announced seed: the forward launch computed these
            return (g_sdf.view(s_sdf), None, g_shape.view(s_shape), None if g_expr is None else g_expr.view(s_expr), None, None, None, None, None)
        from . import _lib
        lib = _lib.load()
        sdf_c, valid_c, zs, ze, obs_idx, thr, lam6 = ctx.saved_tensors
        dev = sdf_c.device
        g_sdf = torch.empty_like(sdf_c)
        g_shape = torch.empty_like(zs)
        g_expr = None if ze is None else torch.empty_like(ze)
        go = g_total.detach().reshape(1).float().contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        n_obs, expr_dim = (ze.shape[0], ze.shape[-1]) if ze is not None else (0, 0)
        _lib.check(lib.nphm_fit_loss_backward(sdf_c.data_ptr(), None if valid_c is None else valid_c.data_ptr(), sdf_c.numel(),
                                              thr.data_ptr(), lam6.data_ptr(), zs.data_ptr(), None if ze is None else ze.data_ptr(),
                                              None if ze is None else obs_idx.data_ptr(), 0 if ze is None else obs_idx.numel(), n_obs,
                                              expr_dim, go.data_ptr(), g_sdf.data_ptr(), g_shape.data_ptr(),
                                              None if g_expr is None else g_expr.data_ptr(), stream), "nphm_fit_loss_backward")
        return (g_sdf.view(s_sdf), None, g_shape.view(s_shape), None if g_expr is None else g_expr.view(s_expr), None, None, None, None, None)


class _GatherRowsFn(torch.autograd.Function):
    """table[idx] along dim 0 (the expression codes of the drawn observations, fitting.py:83) whose backward is ONE launch
    (``nphm_gather_rows_backward``: per row the sum over its draws, in draw order) - autograd's index_put(accumulate)
    sorts the indices first: eight launches for five rows."""

    @staticmethod
    def forward(ctx, table, idx):
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return table.detach().index_select(0, idx)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        g = g.contiguous().float()
        out = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        width = out.numel() // out.shape[0]
        _lib.check(lib.nphm_gather_rows_backward(g.data_ptr(), idx.data_ptr(), idx.numel(), out.shape[0], width, out.data_ptr(),
                                                 torch.cuda.current_stream(g.device).cuda_stream), "nphm_gather_rows_backward")
        return out, None


class _FitInputsFn(torch.autograd.Function):
    """(obs [B,n,C], z_ex [B,1,E], glob_cond [B,1,L+E]) of a step from its draw - the sampled points of the drawn observations,
    their expression codes ``lat_rep[obs_idx]`` and the conditioning rows ``cat([lat_rep_shape on every row, z_ex])``
    (fitting.py:61-85) - in ONE launch (``nphm_fit_inputs``: an advanced-indexing gather, an index_select and a cat in the
    PyTorch formulation), and one launch back (``nphm_fit_inputs_backward``: per code the sum over its draws in draw order,
    straight from a column slice of the conditioning's gradient)."""

    @staticmethod
    def forward(ctx, z_shape, table, drawn, clouds, n_batch, extra=0, sampler=None):
        from . import _lib
        lib = _lib.load()
        dev = table.device
        n_obs, P, C = clouds.shape
        n = (drawn.numel() - n_batch - extra) // n_batch
        L, E = z_shape.shape[-1], table.shape[-1]
        obs = torch.empty(n_batch, n, C, dtype=torch.float32, device=dev)
        z_ex = torch.empty(n_batch, 1, E, dtype=torch.float32, device=dev)
        glob = torch.empty(n_batch, 1, L + E, dtype=torch.float32, device=dev)
        ring = getattr(sampler, "ring", None)
        if ring is not None and ring.shape[1] == drawn.numel() and torch.cuda.is_current_stream_capturing():
            # the recording of a replayed step: the draw comes out of the host ring (and is mirrored into ``drawn``)
            _lib.check(lib.nphm_fit_inputs_ring(ring.data_ptr(), ring.shape[0], ring.stride(0), drawn.numel(), sampler.ring_ctl.data_ptr(),
                                                drawn.data_ptr(), n_batch, n, clouds.data_ptr(), n_obs, P, C, z_shape.detach().data_ptr(), L,
                                                table.detach().data_ptr(), E, obs.data_ptr(), z_ex.data_ptr(), glob.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream), "nphm_fit_inputs_ring")
            sampler.ring_recorded = True
        else:
            _lib.check(lib.nphm_fit_inputs(drawn.data_ptr(), n_batch, n, clouds.data_ptr(), n_obs, P, C, z_shape.detach().data_ptr(), L,
                                           table.detach().data_ptr(), E, obs.data_ptr(), z_ex.data_ptr(), glob.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "nphm_fit_inputs")
        ctx.save_for_backward(drawn)
        ctx.meta = (n_batch, n_obs, L, E, z_shape.shape, table.shape)
        ctx.mark_non_differentiable(obs)
        ctx.set_materialize_grads(False)
        # one alias of the identity code per use in the step (anchor head, identity field, regularisers, compressor) and one of
        # the expression codes (regulariser): same storage and version counter - every cache keyed on them sees ONE tensor -
        # but separate autograd edges, whose gradients the backward launch adds in a fixed order (autograd would run one
        # elementwise add per extra use of a tensor: four launches per step)
        return (obs, z_ex, glob) + tuple(z_shape.detach() for _ in range(4)) + (table.detach(),)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _g_obs, g_z_ex, g_glob, *g_uses):
        from . import _lib
        lib = _lib.load()
        (drawn,) = ctx.saved_tensors
        B, n_obs, L, E, s_shape, s_table = ctx.meta
        g_uses = [None if g is None else g.contiguous().float() for g in g_uses]
        g_shape_uses, g_table_use = g_uses[:4], g_uses[4]
        if g_z_ex is None and g_glob is None and all(g is None for g in g_uses):
            return None, None, None, None, None, None, None
        dev = drawn.device
        stride = 0
        if g_z_ex is not None:
            g_z_ex = g_z_ex.float()
            if g_z_ex.stride(-1) != 1 or (g_z_ex.dim() == 3 and g_z_ex.shape[1] != 1):
                g_z_ex = g_z_ex.contiguous()
            stride = g_z_ex.stride(0)
        if g_glob is not None:
            g_glob = g_glob.contiguous().float()
        for g in g_shape_uses:
            assert g is None or g.numel() == L, "gradient of an identity-code alias"
        assert g_table_use is None or g_table_use.numel() == n_obs * E, "gradient of the expression-code alias"
        g_table = torch.empty(s_table, dtype=torch.float32, device=dev)
        want_shape = g_glob is not None or any(g is not None for g in g_shape_uses)
        g_shape = torch.empty(s_shape, dtype=torch.float32, device=dev) if want_shape else None
        import ctypes
        uses = (ctypes.c_void_p * 4)(*[None if g is None else g.data_ptr() for g in g_shape_uses])
        _lib.check(lib.nphm_fit_inputs_backward(None if g_z_ex is None else g_z_ex.data_ptr(), stride,
                                                None if g_glob is None else g_glob.data_ptr(), drawn.data_ptr(), B, n_obs, L, E,
                                                uses, None if g_table_use is None else g_table_use.data_ptr(),
                                                g_table.data_ptr(), None if g_shape is None else g_shape.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream), "nphm_fit_inputs_backward")
        return g_shape, g_table, None, None, None, None, None


class _StepCodes:
    """the two codes as the step's uses see them: the leaves themselves, or (fused inputs) one alias per use"""

    def __init__(self, lat_rep_shape, lat_rep, shape_uses=None, table_use=None):
        self.anchors, self.field, self.loss, self.compressor = shape_uses if shape_uses is not None else (lat_rep_shape,) * 4
        self.expr_loss = lat_rep if table_use is None else table_use


def _step_inputs(sampler, drawn_dev, lat_rep_shape, lat_rep, n_batch):
    """(obs_idx [B], obs [B,n,C], z_ex [B,1,E], glob_cond [B,1,L+E], _StepCodes) of a joint-fit step (fitting.py:61-85)"""
    import os
    obs_idx = drawn_dev[:n_batch]
    if (sampler.on_gpu and os.environ.get("NPHM_AMD_FIT_FUSED", "1") not in ("0", "") and sampler.clouds.dtype == torch.float32
            and lat_rep.dtype == torch.float32 and lat_rep.is_contiguous() and lat_rep_shape.is_contiguous()
            and lat_rep.dim() == 3 and lat_rep.shape[1] == 1 and lat_rep_shape.shape[:2] == (1, 1) and drawn_dev.is_contiguous()):
        obs, z_ex, glob_cond, *uses = _FitInputsFn.apply(lat_rep_shape, lat_rep, drawn_dev, sampler.clouds, n_batch, sampler.extra, sampler)
        return obs_idx, obs, z_ex, glob_cond, _StepCodes(lat_rep_shape, lat_rep, uses[:4], uses[4])
    obs_idx, obs = sampler.gather(drawn_dev)
    z_ex = _rows_of(lat_rep, obs_idx)
    glob_cond = torch.cat([lat_rep_shape.expand(n_batch, -1, -1), z_ex], dim=-1)
    return obs_idx, obs, z_ex, glob_cond, _StepCodes(lat_rep_shape, lat_rep)


def _rows_of(table, idx):
    """table[idx, ...] (rows along dim 0)"""
    if table.is_cuda and table.dtype == torch.float32 and table.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous():
        return _GatherRowsFn.apply(table, idx)
    return table[idx]


class _ImplicitRootFn(torch.autograd.Function):
    """x_c = root - J^-1 (F(root) - F(root).detach()) (fitting.py:99-106): the value is the root itself, the gradient
    g_posed = -J^-T g_xc flows to the posed points F(root) - one small launch each way instead of the einsum chain."""

    @staticmethod
    def forward(ctx, root, posed, jac_inverse):
        ctx.save_for_backward(jac_inverse)
        return root.detach()               # an alias of the root's storage (nothing writes into either)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_xc):
        from . import _lib
        lib = _lib.load()
        (jinv,) = ctx.saved_tensors
        g = g_xc.contiguous().float()
        out = torch.empty_like(g)
        _lib.check(lib.nphm_fit_root_backward(jinv.data_ptr(), g.data_ptr(), out.data_ptr(), g.numel() // 3,
                                              torch.cuda.current_stream(g.device).cuda_stream), "nphm_fit_root_backward")
        return None, out, None


def _fused_losses_ok(decoder, lambdas, device) -> bool:
    """the fused loss kernels cover the NPHM identity code layout (64 + 40 x 32) and the published loss terms"""
    import os
    if os.environ.get("NPHM_AMD_FIT_FUSED", "1") in ("0", ""):
        return False
    # (fit_loss_kernel hard-codes the split of the code - 64 global + 40 x 32 local columns, the unobserved blocks and the 16
    # symmetric pairs: a decoder with the same total width but another split must not reach it)
    return (device.type == "cuda" and hasattr(decoder, "lat_dim_glob") and getattr(decoder, "backend", "hip") == "hip"
            and getattr(decoder, "lat_dim", 0) == 1344 and getattr(decoder, "num_symm_pairs", 0) == 16
            and getattr(decoder, "lat_dim_glob", 0) == 64 and getattr(decoder, "lat_dim_loc", 0) == 32
            and getattr(decoder, "num_kps", 0) == 39
            and callable(getattr(decoder, "hip_supported", None)) and decoder.hip_supported()
            and all(k in _LOSS_SLOTS for k in lambdas))


def _field_of_one_code(decoder, x, code, cond, local):
    """decoder(x [B,N,3], the ONE identity code on every batch row)[0] (fitting.py:109 / :238).  The field is pointwise and the
    rows share the code, so the NPHM decoder in train mode (no last-point overwrite to reproduce) takes them as one row of
    B * N points: one latent prologue and one latent gradient instead of B, 40 member point lists instead of 40 B (fewer
    partly filled tiles), and no expand / sum of the code's gradient over the rows.  Same values."""
    if local and getattr(decoder, "training", False) and x.is_contiguous() and code.shape[0] == 1:
        sdf, _ = decoder(x.reshape(1, -1, 3), code, None)
        return sdf.reshape(x.shape[0], x.shape[1], 1)
    return decoder(x, cond, None)[0]


def _masked_surface_loss(sdf, thr, valid=None):
    """mean |sdf| over the (valid) points below the clamp ``thr`` (fitting.py:115-132; ``_StepControls`` holds the
    clamp as a device scalar).  The reference compacts the tensor with boolean masks (a device->host sync per mask);
    the same mean as a masked sum / count has static shapes and needs no sync."""
    l = sdf.abs()
    keep = l < thr
    if valid is not None:
        keep = keep & valid.reshape(valid.shape + (1,) * (l.dim() - valid.dim()))
    return torch.where(keep, l, torch.zeros_like(l)).sum() / keep.sum()


class _GraphedStep:
    """Runs ``body()`` eagerly for the first ``warm`` calls (real steps: they consume the RNG and move the
    latents like any other), then records it ONCE into a hipGraph (torch.cuda.CUDAGraph: forward, backward and
    every fused kernel launched on the capture stream) and replays the graph for each later step.  ``body`` reads
    its inputs from static tensors and returns static tensors; the optimizer steps stay outside.  Any failure to
    capture falls back to eager execution for the rest of the loop."""

    def __init__(self, body, enabled, params, warm=3, checks=(), check_every=100):
        self.body, self.enabled, self.params, self.warm = body, bool(enabled), list(params), warm
        self.calls, self.graph, self.out = 0, None, None
        # ``checks``: callables run OUTSIDE the graph every ``check_every`` replays (numerics decisions that were baked into
        # the recording, e.g. DeepSDF.reverify_fit); one returning True drops the recording: the next call records again
        self.checks, self.check_every, self.replays, self.recaptures = list(checks), int(check_every), 0, 0
        self._stale = False

    def zero_grad(self):
        if self.graph is None:                       # once captured, backward REWRITES the static .grad buffers
            for p in self.params:
                p.grad = None

    def eager(self):
        """one step outside the graph (a draw whose shape differs from the graph's static input): after the capture
        backward ACCUMULATES into the graph's static .grad buffers, so they are cleared first"""
        for p in self.params:
            if self.graph is None:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        return self.body()

    def __call__(self):
        if not self.enabled or (self.graph is None and self.calls < self.warm):
            self.calls += 1
            return self.body()
        if self._stale:                              # a check invalidated the recording: backward must ASSIGN fresh .grad
            torch.cuda.synchronize()                 # tensors again while the new graph is recorded (see zero_grad)
            self._stale, self.graph, self.out = False, None, None
            self.recaptures += 1
            for p in self.params:
                p.grad = None
        if self.graph is None:
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.out = self.body()
                self.graph = graph
            except Exception as e:                   # noqa: BLE001 - any capture problem: stay eager
                import warnings
                warnings.warn(f"nphm_amd.fitting: hipGraph capture of the fitting step failed ({e!r}); running eagerly")
                torch.cuda.synchronize()
                self.enabled = False
                for p in self.params:
                    p.grad = None
                return self.body()
        self.graph.replay()
        self.replays += 1
        if self.checks and self.check_every > 0 and self.replays % self.check_every == 0:
            if any([bool(c()) for c in self.checks]):
                self._stale = True                   # record again at the next call (this call's outputs stay valid until then)
        return self.out


def _run_step(step, sampler, drawn_static, drawn_cur, pair=None):
    """Draw this step's sample (host RNG, reference order) and run the step on it.  The graph reads the static index
    tensor; observations of different sizes below n_points can yield a draw of another length (every row of a draw
    has the length of ITS observation, fitting.py:64-70, and the reference needs them equal within a step only): such
    a step runs eagerly on its own index tensor.  ``pair``: the step's optimizer scalars ride behind the indices."""
    drawn = sampler.draw()
    if pair is not None:
        drawn = torch.cat([drawn, pair.host_scalars()])
    sampler.last_seq = None                                # (set by ring_commit when this call turns out to be a ring replay)
    if drawn.shape == drawn_static.shape:
        ring = sampler.ring is not None and getattr(step, "enabled", False)
        if ring and (step.graph is None or step._stale):
            sampler.ring_recorded = sampler.rows_logged = False      # this call may record the step: _FitInputsFn / _FitLossFn say how
        if ring:
            sampler.ring_write(drawn)
        if not (ring and step.graph is not None and not step._stale and sampler.ring_recorded):
            sampler.upload(drawn, out=drawn_static)        # (a pure replay of a ring recording needs none: its first launch reads the ring)
        drawn_cur[0] = drawn_static
        before = getattr(step, "replays", 0)
        out = step()
        if ring and step.replays != before and sampler.ring_recorded:
            sampler.ring_commit()
        return out
    drawn_cur[0] = sampler.upload(drawn)
    try:
        return step.eager()
    finally:
        drawn_cur[0] = drawn_static


def _step_events(timing, step):
    """HIP events around a REPLAYED step (eager steps are host-bound and not what ``timing`` reports)"""
    if timing is None or step.graph is None:
        return None
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ev[0].record()
    timing.setdefault("_events", []).append(ev)
    return ev


def _flush_timing(timing):
    if timing is None:
        return
    evs = timing.pop("_events", [])
    if evs:
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in evs]
        timing["graph_ms"], timing["graph_steps"] = float(sum(ms) / len(ms)), len(ms)


def _graph_default(device, verbose, *decoders):
    """hipGraph replay of the step: on by default for HIP-backed decoders on a ROCm device when nothing has to be
    printed per step (NPHM_AMD_FIT_GRAPH=0 turns it off)."""
    import os
    if os.environ.get("NPHM_AMD_FIT_GRAPH", "1") in ("0", "") or verbose or device.type != "cuda":
        return False
    return all(getattr(d, "backend", "hip") == "hip" for d in decoders if d is not None)


def inference_iterative_root_finding_joint(decoder, decoder_expr, all_obs: List[torch.Tensor], lambdas, n_steps,
                                           schedule_cfg: Dict, step_scale=1, lr_scale=1, *, verbose: bool = True,
                                           history: Optional[list] = None, compute_unused_sdf_grad: bool = False,
                                           use_graph: Optional[bool] = None, timing: Optional[dict] = None):
    """Joint fit of one identity code and one expression code per observation (fitting.py:14-177).
    Returns (lat_rep [n_obs,1,lat_dim_expr], lat_rep_shape [1,1,lat_dim], anchors).

    The reference also evaluates ``nabla(decoder, p_corresp, ...)`` every step (:112) and never uses
    the result; it has no side effect on the fit (no RNG, no parameter, no in-place update), so it is
    skipped unless ``compute_unused_sdf_grad`` — the fitted latents are identical either way.

    Same arithmetic as the reference, arranged without host synchronisation inside a step: conditioning is passed
    as one row per batch entry ([B,1,L] / anchors [B,39,3]: the fields broadcast it; the reference ``repeat``s it per
    point), the sampled points are gathered on the device, the surface loss over the converged correspondences is
    a masked mean, loss weights / the loss clamp are device scalars, the loss trace stays on the device until the
    loop ends (``verbose`` printing reads it every step, like the reference).  On a ROCm device the whole step
    (forward, backward, every fused kernel) is then recorded once into a hipGraph and replayed (``use_graph``).
    ``timing`` (a dict) receives ``graph_ms`` = mean device time of one replayed step between two HIP events on the
    launch stream, and ``graph_steps`` (bench.py: the share of a step the GPU is busy)."""
    device = all_obs[0].device
    n_obs = len(all_obs)
    n_batch, n_points = 5, 1000
    lat_dim_expr = decoder_expr.lat_dim_expr if hasattr(decoder_expr, "lat_dim_expr") else 200
    lat_rep = torch.zeros([n_obs, 1, lat_dim_expr], device=device).float()
    lat_rep.requires_grad = True
    lat_rep_shape = torch.zeros([1, 1, decoder.lat_dim], device=device)
    lat_rep_shape.requires_grad = True
    opt = _adam(lat_rep_shape, 0.01 * lr_scale)
    opt_expr = _adam(lat_rep, 0.01 * lr_scale)
    local = hasattr(decoder, "lat_dim_loc")
    sampler = _ObservationSampler(all_obs, n_batch, n_points)
    n_iter = int(n_steps * step_scale)
    hist = _History(history, lambdas.keys(), n_iter, device, extra=("n_valid",))
    ctl = _StepControls(lambdas, device)
    fused = _fused_losses_ok(decoder, lambdas, device) and "reg_expr" in lambdas
    if fused:
        hist.perm = ctl.fused_perm(extra=("n_valid",))
    if use_graph is None:
        use_graph = _graph_default(device, verbose, decoder, decoder_expr) and not compute_unused_sdf_grad
    import os
    side = None                                              # a second stream for the step's independent branch (see body_in_scope)
    if fused and device.type == "cuda" and os.environ.get("NPHM_AMD_FIT_OVERLAP", "1") not in ("0", ""):
        side = torch.cuda.Stream(device)
    pair = _PairAdam((opt, opt_expr)) if fused else None       # both optimizer steps as one launch inside the step
    if pair is not None and not pair.ok:
        pair = None
    if pair is not None:
        sampler.extra = pair.SLOTS
    drawn_static = sampler.upload(sampler.draw_like())       # static input of the step: the sampled indices (+ the optimizers' scalars)
    drawn_cur = [drawn_static]                               # what the step reads (another tensor for odd-shaped draws)
    if use_graph and fused:
        # replayed steps read their draw from pinned host memory and leave their loss row in a replay-indexed log
        sampler.enable_ring(drawn_static, log_rows=n_iter if hist.buf is not None else 0)
        hist.log = getattr(sampler, "row_log", None)

    def body():
        with (decoder_expr.condition_scope() if hasattr(decoder_expr, "condition_scope") else nullcontext()), \
                (decoder.anchor_scope() if hasattr(decoder, "anchor_scope") else nullcontext()):
            return body_in_scope()

    def body_in_scope():
        obs_idx, obs, z_ex, glob_cond, codes = _step_inputs(sampler, drawn_cur[0], lat_rep_shape, lat_rep, n_batch)     # glob_cond [B,1,L]
        # anchors of the current identity code (the reference runs an N = 1 forward and drops its SDF)
        anchors = _anchors_of(decoder, codes.anchors, device)
        anchors_b = anchors.expand(n_batch, -1, -1) if (local and anchors is not None) else None        # [B,39,3]
        if (side is not None and local and hasattr(decoder, "prefetch_state") and getattr(decoder, "training", False)
                and os.environ.get("NPHM_AMD_FIT_PREFETCH", "1") not in ("0", "")):
            decoder.prefetch_state(codes.field, anchors, side)     # the identity field's prologue, beside the correspondence search

        if hasattr(decoder_expr, "prime_condition"):
            # one conditioning per step, shared by the calls below; its identity half from ONE row
            decoder_expr.prime_condition(glob_cond, anchors_b, parts=(codes.compressor, z_ex, anchors) if anchors_b is not None else None)
        # canonical correspondences by Broyden root finding (no gradient flows through it)
        p_corresp, search_result = search(obs, glob_cond, decoder_expr,
                                          None if anchors_b is None else anchors_b.detach(), multi_corresp=False)
        p_corresp = p_corresp.detach()
        valid = search_result["valid_ids"]

        # implicit differentiation of the root: d x_c = -J^-1 d F(x_c; z) attached to the detached root
        xc = None
        if hasattr(decoder_expr, "implicit_root"):
            if side is not None:
                # The launch pair behind this call (value + Jacobian + inverse + the backward's state at the roots) feeds nothing
                # before the conditioning's backward - x_c's VALUE is the root itself.  On a second stream it runs beside the
                # identity field's forward and backward (neither fills the chip: 313 / 157 / 380 workgroups on 256 CUs);
                # autograd runs its backward on that stream too and orders the two (inside a capture: parallel branches of the graph).
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    xc = decoder_expr.implicit_root(p_corresp, glob_cond, anchors_b)
                if xc is None:
                    torch.cuda.current_stream(device).wait_stream(side)
            else:
                xc = decoder_expr.implicit_root(p_corresp, glob_cond, anchors_b)
        pj = None
        if xc is None and hasattr(decoder_expr, "posed_and_jacobian"):
            pj = decoder_expr.posed_and_jacobian(p_corresp, glob_cond, anchors_b, inverse=True)
        if xc is not None:                 # posed points, Jacobian, its inverse, the state of the backward: one launch; -J^-T in the backward launch
            pass
        elif pj is not None:               # (the same with the implicit function's backward as a launch of its own)
            preds_posed, jac_posed, grad_inv = pj
            xc = _ImplicitRootFn.apply(p_corresp, preds_posed, grad_inv)
        else:
            preds_posed, _ = decoder_expr(p_corresp, glob_cond, anchors_b)
            preds_posed = preds_posed + p_corresp
            jac_posed = jac(decoder_expr, p_corresp, glob_cond, anchors_b)
            grad_inv = _inverse3x3(jac_posed.detach())
            correction = preds_posed - preds_posed.detach()
            # 3x3 matrix-vector products per point, elementwise: as an einsum this is a rocBLAS batched GEMM of 5000 3x3
            # problems (69 us forward + 40 us backward per step)
            correction = -(grad_inv.detach() * correction.unsqueeze(-2)).sum(dim=-1)
            xc = p_corresp + correction

        shape_cond = codes.field.expand(n_batch, -1, -1) if local else codes.field.repeat(n_batch, xc.shape[1], 1)
        sdf = _field_of_one_code(decoder, xc, codes.field, shape_cond, local)
        if compute_unused_sdf_grad:
            _, sdf_grad = nabla(decoder, p_corresp, shape_cond, None)    # dead value in the reference (:112)

        if fused:                          # every loss term, the total and (backward) their gradients: two launches
            loss, row8 = _FitLossFn.apply(sdf, valid, codes.loss, codes.expr_loss, obs_idx, ctl.thr, ctl.lam6, ctl.one,
                                          sampler if sampler.ring is not None else None)
            loss.backward(gradient=ctl.one)       # (a preallocated seed, announced to the loss: its launch computes the gradients too)
            row = row8                     # raw: hist.perm orders it on the host
            if pair is not None:
                pair.launch(drawn_cur[0][drawn_cur[0].numel() - pair.SLOTS:])
        else:
            loss_dict = {"surface": _masked_surface_loss(sdf, ctl.thr, valid),
                         "reg_expr": (torch.norm(codes.expr_loss[obs_idx, :, :], dim=-1) ** 2).mean()}
            _shape_regularisers(decoder, codes.loss, loss_dict)
            loss = ctl.total(loss_dict)
            loss.backward(gradient=ctl.one)
            row = hist.row(loss_dict, loss, n_valid=valid.sum())
        return row, anchors.detach()

    # the expression decoder's fitting launches run a two-term layer mask that was measured on the FIRST codes and is baked
    # into the recording: it is measured again on the current codes every fit_verify_every steps (DeepSDF.reverify_fit)
    mlp = getattr(decoder_expr, "defDeepSDF", None)
    checks = [mlp.reverify_fit] if (mlp is not None and hasattr(mlp, "reverify_fit")) else []
    step = _GraphedStep(body, use_graph, [lat_rep_shape, lat_rep], checks=checks,
                        check_every=getattr(mlp, "fit_verify_every", 100) if mlp is not None else 0)
    anchors = None
    done = 0
    with _frozen(decoder, decoder_expr):
        for j in range(n_iter):
            _apply_schedule(j, step_scale, schedule_cfg, lambdas, (opt, opt_expr), True)
            ctl.refresh(lambdas, j, step_scale)
            step.zero_grad()
            ev = _step_events(timing, step)
            row, anchors = _run_step(step, sampler, drawn_static, drawn_cur, pair)
            if ev is not None:
                ev[1].record()
            if pair is None:
                opt.step()
                opt_expr.step()
            else:
                pair.bump()
            if sampler.ring is not None and sampler.last_seq is not None and sampler.rows_logged and hist.log is not None:
                hist.note(j, sampler.last_seq)                 # (the replay's loss launch stored the row itself)
            else:
                hist.record(j, row)
            done = j + 1
            if verbose:
                r = hist.ordered(row.cpu().numpy())
                _report(j, lambdas, dict(zip(ctl.keys, r)), int(round(float(r[-1]))))
    hist.flush(done)
    _flush_timing(timing)
    if anchors is not None:
        anchors = anchors.clone()                              # out of the graph's memory pool

    return lat_rep, lat_rep_shape, anchors


def inference_identity_space(decoder, all_obs: List[torch.Tensor], lambdas, n_steps, schedule_cfg: Dict,
                             step_scale=1, lr_scale=1, *, verbose: bool = False, history: Optional[list] = None,
                             use_graph: Optional[bool] = None):
    """Identity-only fit on neutral observations (fitting.py:180-288): no deformation field, the
    observed points are canonical points.  Returns (lat_rep_shape, anchors)."""
    device = all_obs[0].device
    n_batch, n_points = 5, 1000
    lat_rep_shape = torch.zeros([1, 1, decoder.lat_dim], device=device)
    lat_rep_shape.requires_grad = True
    opt = _adam(lat_rep_shape, 0.01 * lr_scale)
    local = hasattr(decoder, "lat_dim_loc")
    sampler = _ObservationSampler(all_obs, n_batch, n_points)
    n_iter = int(n_steps * step_scale)
    hist = _History(history, lambdas.keys(), n_iter, device)
    ctl = _StepControls(lambdas, device)
    fused = _fused_losses_ok(decoder, lambdas, device) and "reg_expr" not in lambdas
    if fused:
        hist.perm = ctl.fused_perm()
    if use_graph is None:
        use_graph = _graph_default(device, verbose, decoder)
    pair = _PairAdam((opt,)) if fused else None             # the optimizer step as a launch inside the step (see the joint loop)
    if pair is not None and not pair.ok:
        pair = None
    if pair is not None:
        sampler.extra = pair.SLOTS
    drawn_static = sampler.upload(sampler.draw_like())
    drawn_cur = [drawn_static]

    def body():
        anchors = _anchors_of(decoder, lat_rep_shape, device)
        _, obs = sampler.gather(drawn_cur[0])
        cond = lat_rep_shape.expand(n_batch, -1, -1) if local else lat_rep_shape.repeat(n_batch, obs.shape[1], 1)
        sdf = _field_of_one_code(decoder, obs, lat_rep_shape, cond, local)
        if fused:
            loss, row8 = _FitLossFn.apply(sdf, None, lat_rep_shape, None, None, ctl.thr, ctl.lam6, ctl.one)
            loss.backward(gradient=ctl.one)       # (the announced seed: the loss launch wrote the gradients too)
            if pair is not None:
                pair.launch(drawn_cur[0][drawn_cur[0].numel() - pair.SLOTS:])
            return row8, anchors.detach()
        loss_dict = {"surface": _masked_surface_loss(sdf, ctl.thr)}
        _shape_regularisers(decoder, lat_rep_shape, loss_dict)
        loss = ctl.total(loss_dict)
        loss.backward(gradient=ctl.one)
        return hist.row(loss_dict, loss), anchors.detach()

    step = _GraphedStep(body, use_graph, [lat_rep_shape])
    anchors = None
    done = 0
    with _frozen(decoder):
        for j in range(n_iter):
            _apply_schedule(j, step_scale, schedule_cfg, lambdas, (opt,), False)
            ctl.refresh(lambdas, j, step_scale)
            step.zero_grad()
            row, anchors = _run_step(step, sampler, drawn_static, drawn_cur, pair)
            if pair is None:
                opt.step()
            else:
                pair.bump()
            hist.record(j, row)
            done = j + 1
            if verbose:
                _report(j, lambdas, dict(zip(ctl.keys, hist.ordered(row.cpu().numpy()))))
    hist.flush(done)
    if anchors is not None:
        anchors = anchors.clone()

    return lat_rep_shape, anchors
