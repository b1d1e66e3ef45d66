#!/bin/bash
# usage (GPU box): tools/ab_bench.sh lib1.so lib2.so ...  - bench.py value of each library variant, twice, interleaved
for rep in 1 2; do for l in "$@"; do
  printf "%s " "$l"; NPHM_AMD_LIB=$PWD/$l python bench.py --no-cpu-baseline --steps 8 ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],2))"
done; done
