#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]
# Results land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-sub --no-mesh $*"
echo "cmd: $CMD" > "$OUT/README.txt"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- $CMD > "$OUT/trace.log" 2>&1
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d "$OUT/$name" -o $name --output-format csv -- $CMD > "$OUT/$name.log" 2>&1
}
pass pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS
pass pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
find "$OUT" -name "*.csv" | head -40 >> "$OUT/README.txt"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in sorted(glob.glob(out + "/**/*kernel_stats.csv", recursive=True)):
    lines.append("== " + os.path.relpath(f, out))
    lines += open(f).read().splitlines()[:12]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    lines.append("== " + os.path.relpath(f, out) + " (sum over dispatches; n = dispatches)")
    for k, d in agg.items():
        for c, v in d.items():
            lines.append(f"{k:60s} {c:28s} sum={v:.6g} n={cnt[(k, c)]} per_dispatch={v / cnt[(k, c)]:.6g}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
