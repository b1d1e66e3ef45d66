#!/usr/bin/env python3
"""Host marching cubes timing at 256^3 on this box's cores (thread sweep).  Development tool."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nphm_amd import reconstruction as R  # noqa: E402

res = 256
ax = np.linspace(-0.5, 0.5, res, dtype=np.float32)
g = (np.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2) - 0.35).astype(np.float32)
print("cores", os.cpu_count())
for th in (1, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        v, f = R.marching_cubes(g, 0.0, negate=True, n_threads=th)
        best = min(best, time.perf_counter() - t)
    print(f"{th:4d} threads {best * 1e3:8.1f} ms  {v.shape} {f.shape}")
t = time.perf_counter(); h = g.copy(); h *= -1; print("numpy negate", (time.perf_counter() - t) * 1e3, "ms")
