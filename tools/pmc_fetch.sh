#!/bin/bash
# Run on the GPU box: FETCH_SIZE of bench.py's kernels for the library in $NPHM_AMD_LIB (one counter, one
# pass, under a timeout: a three-counter pass FETCH_SIZE + TCC_HIT_sum + TCC_MISS_sum hung the profiler).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
OUT=$ROOT/gpurun_out/prof_fetch_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-mesh"
timeout 150 rocprofv3 --pmc FETCH_SIZE -d "$OUT/f" -o f --output-format csv -- $CMD > "$OUT/f.log" 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(sys.argv[2], {k: round(acc[k] / max(1, n[k])) for k in acc})
PY
