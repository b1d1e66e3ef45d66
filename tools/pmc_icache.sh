#!/bin/bash
# Run on the GPU box: instruction-cache counters of bench.py's kernels (own PMC passes, no tracing).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_icache
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH -d "$OUT/ic1" -o ic1 --output-format csv -- $CMD > "$OUT/ic1.log" 2>&1
rocprofv3 --pmc InstrFetchLatency -d "$OUT/ic2" -o ic2 --output-format csv -- $CMD > "$OUT/ic2.log" 2>&1
rocprofv3 --pmc SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$OUT/ic3" -o ic3 --output-format csv -- $CMD > "$OUT/ic3.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "eval_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f.split("/")[-2], k, "per dispatch", acc[k] / max(1, n[k]), "n", n[k])
PY
