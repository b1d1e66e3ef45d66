#!/bin/bash
# usage (GPU box): tools/power_ab.sh lib1.so lib2.so ...   - for every library variant: the identity lattice kernel back to back for
# a few seconds (bench.py --workload identity --steps 300) with rocm-smi socket power / shader clock sampled meanwhile.
# One line per variant: Mpoints/s, ms per step, median power, median clock.  Evidence for what limits the identity kernel:
# a timing ablation that removes waiting (no barrier, no vmcnt) and lands at the SAME step time with a LOWER clock is a kernel
# at the chip's power / current limit, not one that waits (tools/build_variant.sh -DNPHM_ABLATE=...).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/power_ab
mkdir -p "$OUT"
cd "$ROOT"
for l in "$@"; do
  name=$(basename "$l" .so)
  NPHM_AMD_LIB=$PWD/$l python bench.py --workload identity --no-cpu-baseline --no-sub --no-mesh --steps ${STEPS:-300} --warmup 5 ${BENCH_ARGS:-} > "$OUT/$name.json" 2> "$OUT/$name.err" &
  pid=$!
  sleep 7                                   # import + calibration + warm-up
  : > "$OUT/$name.smi"
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> "$OUT/$name.smi"
  done
  wait $pid
  python - "$OUT/$name" <<'PY'
import re, sys, json, statistics as st
base = sys.argv[1]
txt = open(base + ".smi").read()
pw = [float(x) for x in re.findall(r"Power \(W\):\s*([0-9.]+)", txt)]
ck = [float(x) for x in re.findall(r"sclk clock level:?\s*\d*:?\s*\(?([0-9.]+)Mhz", txt)]
try:
    d = json.loads(open(base + ".json").read().strip().splitlines()[-1])
    v, ms = d["value"], d["ms_per_step"]
except Exception as e:
    v, ms = float("nan"), float("nan")
med = lambda x: st.median(x) if x else float("nan")
# drop the ramp: samples below 80 % of the median belong to idle gaps
pw2 = [p for p in pw if p > 0.8 * med(pw)]; ck2 = [c for c in ck if c > 0.8 * med(ck)]
print(f"{base.split('/')[-1]:12s} {v:8.1f} Mpoints/s {ms:7.2f} ms/step   power W median {med(pw2):6.0f} (n {len(pw2)})   sclk MHz median {med(ck2):6.0f}   ms x GHz {ms * med(ck2) / 1e3:7.2f}")
PY
done
