#!/usr/bin/env python3
"""How far do (a) the reference's arithmetic on PyTorch-ROCm (composite tier) and (b) the HIP tier land from the reference's
CPU trace over the 250 steps of tests/golden/fitting_long.npz?  Prints the deviation measures tests/test_fitting.py asserts.
(development tool; the test is test_long_horizon_hip_tier_against_the_reference_arithmetic_on_this_gpu)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fitting as T  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    for name, backend, kw in (("composite", "composite", dict(use_graph=False)), ("hip_graph", None, dict(use_graph=True)),
                              ("hip_eager", None, dict(use_graph=False)), ("hip_f16x3", None, dict(use_graph=True, fit_numerics="f16x3"))):
        g, keys, table, lat_e, lat_s, anc = T._run_long(dev, backend, **kw)
        out[name] = T.long_deviation(g, keys, table, lat_e, lat_s, anc)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
