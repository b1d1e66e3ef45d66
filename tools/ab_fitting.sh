#!/bin/bash
# usage (GPU box): tools/ab_fitting.sh OUTDIR  - same-box A/B of the fitting step's launch shapes (dev tool):
# value+Jacobian launches cut 16+8 points (NPHM_AMD_JVP_SPLIT) x conditioning-backward workgroups of 64 / 32 points
# (NPHM_AMD_MLP_BWD_POINTS), interleaved twice; then the trained-pair trace test three times per backward shape
# (its first-steps band is sensitive to summation order: see tests/test_fitting.py)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; out=${1:-gpurun_out/ab_fit}; mkdir -p $out
for rep in 1 2; do for s in 1 0; do for b in 64 32; do
  v=$(NPHM_AMD_JVP_SPLIT=$s NPHM_AMD_MLP_BWD_POINTS=$b python bench.py --workload fitting --no-cpu-baseline --no-sub 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))")
  echo "split=$s bwd=$b steps/s $v" | tee -a $out/ab.txt
done; done; done
for b in 64 32 64 32 64 32; do
  NPHM_AMD_MLP_BWD_POINTS=$b python -m pytest tests/test_fitting.py -m gpu -q -s -k trained_identity_and_deformation 2>&1 | grep "trained pair\|passed\|failed" | sed "s/^/bwd=$b /" | tee -a $out/ab.txt
done
