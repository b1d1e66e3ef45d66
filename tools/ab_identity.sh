#!/bin/bash
# usage (GPU box): tools/ab_identity.sh lib1.so lib2.so ...  - per library variant: (1) parity of a 24^3 lattice against the numpy
# oracle in the default numerics (must stay < 1e-5), (2) bench.py --workload identity value, interleaved, REPS times
# (dev tool for same-box A/B runs; the variants come from tools/build_variant.sh)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for l in "$@"; do
  printf "%s parity: " "$l"
  NPHM_AMD_LIB=$PWD/$l python - <<'PY' 2>&1 | tail -1
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _util as U
from oracle import nphm_oracle as O
from nphm_amd import reconstruction as R
dev = torch.device("cuda:0")
net = U.build_identity(device=dev).eval()
lat = U.sample_latent(0).to(dev)
res = 24
grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).to(dev, dtype=torch.float)[None]
params, amean = U.np_state(net), U.anchors_mean()
fwd = lambda p, l: O.nphm_identity_forward(params, amean, p, l, training=False)
ref = O.get_logits(fwd, lat.cpu().numpy(), grid.cpu().numpy(), nbatch_points=1000)
errs = []
for prec in ("auto", "f16x3", "f16x3a2", "bf16x3", "bf16x3a2", "f32"):
    if prec != "auto": net.precision = prec
    vol = R.get_logits(net, lat, grid, nbatch_points=1000)
    errs.append(f"{prec} {float(np.max(np.abs(vol - ref))):.2e}")
print(" | ".join(errs))
PY
done
for rep in $(seq 1 ${REPS:-2}); do for l in "$@"; do
  printf "%s " "$l"; NPHM_AMD_LIB=$PWD/$l python bench.py --workload identity --no-cpu-baseline --no-sub --no-mesh --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2), d['roofline'].get('frac'))"
done; done
