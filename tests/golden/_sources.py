"""Inputs of the trained-like fixture generators: the training runs' full outputs when they are still around (gpurun_out/ is
scratch, git-ignored), else the COMMITTED copies of the same tensors in this directory - so that every generator runs from a
clean clone (build container, needs /root/reference).

    heads : tools/train_synthetic_heads.py        -> gpurun_out/r3a/trained_heads.npz | trained_state.npz
    expr  : tools/train_synthetic_expressions.py  -> gpurun_out/r4/trained_expr.npz   | trained_def_state.npz + trained_expr_codes.npz
    npm   : tools/train_synthetic_npm.py          -> gpurun_out/r4/trained_npm.npz    | trained_npm_state.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SCRATCH = {"heads": os.path.join(ROOT, "gpurun_out", "r3a", "trained_heads.npz"),
           "expr": os.path.join(ROOT, "gpurun_out", "r4", "trained_expr.npz"),
           "npm": os.path.join(ROOT, "gpurun_out", "r4", "trained_npm.npz")}


def load(kind):
    """-> (dict of arrays with the keys of the training run's output, from_scratch: bool).  argv[1] overrides the path."""
    path = sys.argv[1] if len(sys.argv) > 1 else SCRATCH[kind]
    if os.path.exists(path):
        ck = np.load(path)
        return {k: ck[k] for k in ck.files}, True
    print(f"{path} not found: reading the committed copies under tests/golden/", flush=True)
    if kind == "heads":
        ck = np.load(os.path.join(HERE, "trained_state.npz"))
        d = {k: ck[k] for k in ck.files}
        d["anchors"] = d.pop("subject_anchors")
        return d, False
    if kind == "expr":
        ck = np.load(os.path.join(HERE, "trained_def_state.npz"))
        d = {k: ck[k] for k in ck.files if k.startswith("sd.") or k in ("trace", "meta")}
        d["z_ex"] = np.load(os.path.join(HERE, "trained_expr_codes.npz"))["z_ex"]
        return d, False
    if kind == "npm":
        ck = np.load(os.path.join(HERE, "trained_npm_state.npz"))
        d = {k: ck[k] for k in ck.files if k.startswith("sd.") or k == "meta"}
        ids = [int(i) for i in ck["code_ids"]]
        codes = np.zeros((max(ids) + 1, ck["codes"].shape[1]), dtype=np.float32)
        codes[ids] = ck["codes"]                      # (only the rows the fixture uses are kept in the committed file)
        d["codes"] = codes
        return d, False
    raise KeyError(kind)
