"""CPU tests of the C-ABI boundary: the library loads without a GPU and exports exactly the symbols
include/nphm_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import _util as U
from nphm_amd import _lib


def _header_functions():
    text = open(os.path.join(U.ROOT, "include", "nphm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nphm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _header_functions()
    assert declared, "no functions parsed from the header"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/nphm_amd.h but not exported"
    # and the Python binding table covers the whole header
    assert sorted(_lib.SYMBOLS) == declared


def test_abi_queries_without_gpu():
    lib = _lib.load()
    assert lib.nphm_abi_version() == 12
    assert lib.nphm_identity_supported(64, 32, 39, 16, 200, 4, 1, 3) == 1
    assert lib.nphm_identity_supported(64, 32, 39, 16, 256, 4, 1, 3) == 0
    assert lib.nphm_identity_packed_bytes() > 24 * 81_000 * 4          # >= folded fp32 weights
    assert lib.nphm_identity_latent_state_bytes(3) == 3 * lib.nphm_identity_latent_state_bytes(1)


def test_argument_validation_is_loud():
    lib = _lib.load()
    rc = lib.nphm_identity_eval_points(None, None, None, 1, 10, 0, 0.0, 0, None, None, None)
    assert rc != 0 and b"null pointer" in lib.nphm_last_error()


def test_grid_workspace_size_query():
    """Host-side size query of the binned grid traversal: covers the per-tile and per-point arrays,
    grows with the slab, rejects empty slabs (no GPU needed)."""
    lib = _lib.load()
    def tiles(nx, ny, nz):
        return ((nx + 3) // 4) * ((ny + 3) // 4) * ((nz + 1) // 2)
    for dims in [(3, 5, 1), (37, 26, 19), (64, 64, 64), (256, 256, 256)]:
        need = lib.nphm_identity_grid_workspace_bytes(*dims)
        assert need >= tiles(*dims) * (4 + 4 + 8 + 8 + 16 + 32 * 8)
    assert lib.nphm_identity_grid_workspace_bytes(0, 4, 4) == 0
    assert lib.nphm_identity_grid_workspace_bytes(32, 256, 256) < lib.nphm_identity_grid_workspace_bytes(256, 256, 256)


def test_to_host_passes_cpu_tensors_through():
    import torch
    from nphm_amd import reconstruction as R
    t = torch.arange(6, dtype=torch.float32).reshape(2, 3)
    a = R.to_host(t)
    assert a.shape == (2, 3) and a.dtype.name == "float32" and float(a[1, 2]) == 5.0
