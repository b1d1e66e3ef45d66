"""ctypes binding of libnphm_amd.so (C ABI declared in include/nphm_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``nphm_amd.build``; there is no
fallback: if it is missing or fails to load, every HIP-backed entry point raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NPHM_AMD_LIB") or os.path.join(_HERE, "libnphm_amd.so")   # override: dev builds

NPHM_PREC_F32 = 0
NPHM_PREC_BF16X3 = 1
NPHM_PREC_BF16X3_ADAPTIVE = 2
NPHM_PREC_BF16X3_ADAPTIVE2 = 3
NPHM_PREC_F16X3 = 4
NPHM_PREC_F16X3_ADAPTIVE2 = 5

_PtrArr5 = c_void_p * 5
_PtrArr3 = c_void_p * 3

# name -> (restype, argtypes); must list every symbol of include/nphm_amd.h
SYMBOLS = {
    "nphm_abi_version": (c_int, []),
    "nphm_last_error": (c_char_p, []),
    "nphm_probe_mfma_rate": (c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_void_p]),
    "nphm_identity_supported": (c_int, [c_int] * 8),
    "nphm_identity_packed_bytes": (c_size_t, []),
    "nphm_identity_latent_state_bytes": (c_size_t, [c_int]),
    "nphm_identity_pack": (c_int, [_PtrArr5, _PtrArr5, c_void_p, c_void_p]),
    "nphm_identity_prepare_latent": (c_int, [c_void_p, _PtrArr5, _PtrArr5, _PtrArr3, _PtrArr3, c_int,
                                             c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_prepare_latent_anchors": (c_int, [_PtrArr5, _PtrArr5, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nphm_head_forward": (c_int, [_PtrArr3, _PtrArr3, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nphm_head_backward": (c_int, [_PtrArr3, _PtrArr3, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "nphm_compress_condition": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nphm_compress_condition_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nphm_fit_loss": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                              c_void_p, c_void_p]),
    "nphm_fit_loss_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_fit_loss_with_gradients": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                             c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_fit_loss_with_gradients_logged": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nphm_fit_root_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "nphm_fit_inputs": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "nphm_fit_inputs_ring": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                     c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_fit_inputs_backward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_void_p * 4, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "nphm_identity_latent_grad_scratch_bytes": (c_size_t, [c_int]),
    "nphm_identity_latent_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_set_member_bounds": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "nphm_identity_eval_points": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int64, c_float,
                                          c_int, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_eval_grid": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                        c_int, c_int, c_int, c_int64, c_float, c_int, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]),
    "nphm_identity_grid_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nphm_identity_eval_grid_planes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                               c_int, c_void_p, c_int, c_int64, c_float, c_int, c_void_p, c_void_p,
                                               c_void_p, c_size_t, c_void_p]),
    "nphm_identity_bwd_packed_bytes": (c_size_t, []),
    "nphm_identity_pack_bwd": (c_int, [_PtrArr5, c_void_p, c_void_p]),
    "nphm_identity_list_tiles": (c_int, [c_int64]),
    "nphm_identity_build_lists": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "nphm_identity_member_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int,
                                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_backward_scratch_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "nphm_identity_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                       c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_train_saved_bytes": (c_size_t, [c_int, c_int]),
    "nphm_identity_train_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "nphm_identity_train_edge_bytes": (c_size_t, [c_int]),
    "nphm_identity_train_pair_counts": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "nphm_identity_train_point_list": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_train_tables": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "nphm_identity_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nphm_identity_train_operand_scales": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_train_weight_grads": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nphm_identity_train_wpart_bytes": (c_size_t, [c_int]),
    "nphm_identity_train_reduce_grads": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                                 _PtrArr5, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_blend_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_blend_partial_bytes": (c_size_t, [c_int, c_int64]),
    "nphm_identity_blend_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_identity_eval_grid_points": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_int64, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                               c_void_p]),
    "nphm_mlp_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_int]),
    "nphm_mlp_packed_bytes": (c_size_t, [c_int] * 4),
    "nphm_mlp_latent_state_bytes": (c_size_t, [c_int] * 5),
    "nphm_mlp_pack": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_mlp_prepare_latent": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nphm_mlp_eval_points": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                                    c_void_p, c_void_p]),
    "nphm_mlp_eval_workspace_bytes": (c_size_t, []),
    "nphm_mlp_eval_points_ws": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int,
                                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "nphm_mlp_eval_grid_ws": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                                     c_void_p]),
    "nphm_mlp_eval_points_jvp": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                                        c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "nphm_mlp_broyden": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                                                c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nphm_mlp_broyden_from": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                                     c_int64, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nphm_mlp_saved_bytes": (c_size_t, [c_int] * 5 + [c_int64]),
    "nphm_mlp_eval_points_saving": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                                           c_void_p, c_void_p, c_int, c_void_p]),
    "nphm_mlp_eval_points_jvp_saving": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                                               c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "nphm_mlp_bwd_packed_bytes": (c_size_t, [c_int] * 4),
    "nphm_mlp_pack_bwd": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p]),
    "nphm_mlp_bwd_partial_bytes": (c_size_t, [c_int, c_int, c_int64]),
    "nphm_mlp_backward_cond": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "nphm_inverse3x3": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nphm_train_loss_blocks": (c_int, []),
    "nphm_train_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_train_loss_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nphm_adam_step_pair": (c_int, [c_void_p * 2, c_void_p * 2, c_void_p * 2, c_void_p * 2, c_int64 * 2, c_void_p, c_void_p]),
    "nphm_gather_rows_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nphm_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                               c_void_p]),
    "nphm_identity_blend_members": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "nphm_inverse3x3_strided": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "nphm_mlp_cond_grad": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                   c_void_p, c_void_p]),
    "nphm_mlp_eval_grid": (c_int, [c_int] * 4 + [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nphm_mc_extract": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_double, c_int, c_int,
                                ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "nphm_mc_fetch": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nphm_mc_free": (None, [c_void_p]),
    "nphm_mc_device_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "nphm_mc_device_count": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_double, c_int, c_void_p,
                                     ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), c_void_p]),
    "nphm_mc_device_emit": (c_int, [c_void_p, c_int, c_int, c_int, ctypes.c_double, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "nphm_dense_gemm_nt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_float,
                                   c_int, c_int, c_void_p]),
    "nphm_dense_reduce_splits": (c_int, [c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p]),
    "nphm_dense_gpre": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nphm_dense_column_sums": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


class NphmAmdError(RuntimeError):
    pass


def load():
    """Load libnphm_amd.so (once).  Raises NphmAmdError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NphmAmdError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  nphm_amd has no non-HIP fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.nphm_abi_version() != 12:
        raise NphmAmdError("libnphm_amd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().nphm_last_error()
        raise NphmAmdError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr_array5(tensors):
    return _PtrArr5(*[t.data_ptr() for t in tensors])


def ptr_array(tensors):
    """ctypes array of device pointers (any length)."""
    return (c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def ptr_array3(tensors):
    return _PtrArr3(*[t.data_ptr() for t in tensors])
