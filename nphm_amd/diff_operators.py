"""Autograd operators the fitting loop applies to the fields — host-side mirror of
src/NPHM/models/diff_operators.py (``jac`` :26-54, ``gradient`` :69-79).  They are callers of the
hot path: the fields' differentiable (composite) tier serves them."""
from __future__ import annotations

import torch


def jac(decoder_expr, xc, cond, anchors):
    """Jacobian of the posed position x_d = x_c + F_ex(x_c) w.r.t. the canonical point
    (diff_operators.py:26-54): xc [B,N,3] -> [B,N,3,3] with [..., i, :] = d x_d[i] / d x_c.
    Like the reference it switches ``requires_grad`` on for ``xc`` in place; the result carries no
    graph (the reference's VJPs use create_graph=False).  On a ROCm device the deformation field's
    fused value+Jacobian kernel serves it; otherwise one vector-Jacobian product per output
    coordinate."""
    xc.requires_grad_(True)
    fused = decoder_expr.jacobian(xc, cond, anchors) if hasattr(decoder_expr, "jacobian") else None
    if fused is not None:                       # analytic forward-mode Jacobian, one HIP launch
        return fused[1]
    offsets, _ = decoder_expr(xc, cond, anchors)
    xd = xc + offsets
    rows = []
    for i in range(xd.shape[-1]):
        seed = torch.zeros_like(xd)
        seed[..., i] = 1
        rows.append(torch.autograd.grad(outputs=xd, inputs=xc, grad_outputs=seed, create_graph=False,
                                        retain_graph=True, only_inputs=True)[0])
    return torch.stack(rows, dim=-2)


def gradient(outputs, inputs):
    """d outputs / d inputs[..., -3:] with an all-ones cotangent, graph retained and extended
    (diff_operators.py:69-79) — the SDF normal direction used by the losses."""
    ones = torch.ones_like(outputs)
    g = torch.autograd.grad(outputs=outputs, inputs=inputs, grad_outputs=ones, create_graph=True,
                            retain_graph=True, only_inputs=True, allow_unused=True)[0]
    return g[:, :, -3:]


def _single_stride(J):
    """stride between consecutive 3x3 matrices if the leading dimensions of J [..., 3, 3] collapse into one, else None"""
    if J.dim() < 2 or tuple(J.shape[-2:]) != (3, 3):
        return None
    step = None
    for size, stride in zip(reversed(J.shape[:-2]), reversed(J.stride()[:-2])):
        if size == 1:
            continue
        if step is None:
            step, span = stride, stride * size
        elif stride == span:
            span = stride * size
        else:
            return None
    return 9 if step is None else step


def inverse3x3(J):
    """Batched 3x3 inverse of a CONSTANT: the result never carries a graph (both callers - the root finder's initial
    inverse Jacobian and the implicit-differentiation correction of the fitting loop - detach it; the reference's
    ``Tensor.inverse()`` would be differentiable).  A Jacobian that requires grad is refused rather than silently cut
    from the graph.  The reference's inverse (= linalg.inv) is an LU factorisation followed by a blocking read of
    its error flag; on a ROCm device the adjugate formula runs in one small kernel (``nphm_inverse3x3``: no host sync,
    no library workspace - capturable in a hipGraph; a singular matrix yields inf / nan entries instead of the
    reference's exception), elsewhere inv_ex."""
    if J.requires_grad:
        raise RuntimeError("nphm_amd.diff_operators.inverse3x3 is not differentiable: pass J.detach() "
                           "(use torch.linalg.inv for a differentiable inverse)")
    if J.is_cuda and J.dtype == torch.float32:
        from . import _lib
        lib = _lib.load()
        Jd = J.detach()
        out = torch.empty(Jd.shape, dtype=torch.float32, device=J.device)
        stream = torch.cuda.current_stream(J.device).cuda_stream
        lead = _single_stride(Jd)
        if lead is None:
            Jd, lead = Jd.contiguous(), 9
        # one kernel for contiguous matrices and for transposed / sliced views (the Jacobian block of a value+Jacobian
        # output is inverted where it lies): the same arithmetic, bit for bit
        _lib.check(lib.nphm_inverse3x3_strided(Jd.data_ptr(), lead, Jd.stride(-2), Jd.stride(-1), out.data_ptr(),
                                               Jd.numel() // 9, stream), "nphm_inverse3x3_strided")
        return out
    return torch.linalg.inv_ex(J)[0]
