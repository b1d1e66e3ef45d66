"""TEST INFRASTRUCTURE (not product): import the reference's own modules from the bytecode oracle/build_ref.py left under
oracle/_ref/ (see there).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this."""
import importlib.machinery
import importlib.util
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
_cache = {}


def available() -> bool:
    tag = os.path.join(REF_DIR, "PYTHON")
    if not os.path.exists(tag):
        return False
    return open(tag).read().strip() == "%d.%d" % sys.version_info[:2]


def load(name: str):
    """name in {'EnsembledDeepSDF', 'deepSDF', 'reconstruction'} -> the reference module"""
    if name in _cache:
        return _cache[name]
    if not available():
        raise ImportError("oracle/_ref is missing or was built by another Python version (run oracle/build_ref.py where "
                          "/root/reference exists)")
    # trimesh / mcubes: imported by the reference's reconstruction.py, unused by get_logits* (as tests/golden/make_golden.py).
    # Absent modules are stubbed for the duration of the import ONLY - a stub left in sys.modules would be what the
    # product's own optional `import trimesh` finds afterwards
    stubbed = []
    for missing in ("trimesh", "mcubes"):
        if missing not in sys.modules and importlib.util.find_spec(missing) is None:
            sys.modules[missing] = types.ModuleType(missing)
            stubbed.append(missing)
    try:
        path = os.path.join(REF_DIR, f"{name}.pyc")
        loader = importlib.machinery.SourcelessFileLoader(f"_nphm_reference.{name}", path)
        spec = importlib.util.spec_from_loader(loader.name, loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
    finally:
        for m in stubbed:
            sys.modules.pop(m, None)
    _cache[name] = mod
    return mod


def build_identity(anchors, state_dict=None, pos_mlp_dim=256):
    """The reference's FastEnsembleDeepSDFMirrored with the nphm.yaml architecture (EnsembledDeepSDF.py:153-200);
    ``anchors`` [1,1,39,3] torch float.  ``state_dict``: weights to load (strict)."""
    import contextlib
    import io
    m = load("EnsembledDeepSDF")
    with contextlib.redirect_stdout(io.StringIO()):
        net = m.FastEnsembleDeepSDFMirrored(lat_dim_glob=64, lat_dim_loc=32, n_loc=39, n_symm_pairs=16, anchors=anchors,
                                            hidden_dim=200, n_layers=4, pos_mlp_dim=pos_mlp_dim)
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    return net


def build_npm(state_dict=None):
    import contextlib
    import io
    m = load("deepSDF")
    with contextlib.redirect_stdout(io.StringIO()):
        net = m.DeepSDF(lat_dim=512, hidden_dim=1024, nlayers=8, geometric_init=True)
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    return net
