#!/bin/bash
# Run on the GPU box (via gpurun): HBM-side bytes per launch of bench.py's dominant kernels -> gpurun_out/traffic.json
# (copy it to profiles/traffic.json: bench.py's `roofline.traffic` reads the committed file and says so in
# `traffic_source`).  Recipe of MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains
# next to --pmc), KB units, FETCH_SIZE doubled on gfx950 (64-B request granularity counted as 32), per dispatch.
# usage: tools/pmc_traffic.sh [identity|two_stage|npm ...]   (default: all three)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
WL=${*:-identity two_stage npm}
for w in $WL; do
  CMD="python $ROOT/bench.py --workload $w --no-cpu-baseline --no-sub --no-mesh --steps 2 --warmup 1"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c -d "$OUT/${w}_$c" -o p --output-format csv -- $CMD > "$OUT/${w}_$c.log" 2>&1
  done
done
python - "$OUT" "$ROOT/gpurun_out/traffic.json" <<'PY'
import collections, csv, glob, json, os, sys, time
out, dst = sys.argv[1], sys.argv[2]
pts = {"nphm::eval_kernel<2,2>": 256 ** 3, "nphm::eval_kernel<2,1>": 256 ** 3,
       # dense MLP lattice launches (bench.py: _mlp_kernel_name): <MT,NTW,MODE,KIND (5: last hidden layer two-term in two point halves),F16,no wl fragments,single-term>
       "nphm::mlp::mlp_eval_kernel<4,2,1,0,true,false,true>": 256 ** 3, "nphm::mlp::mlp_eval_kernel<2,2,1,0,true,true,false>": 256 ** 3,
       "nphm::mlp::mlp_eval_kernel<2,2,1,0,true,false,false>": 256 ** 3, "nphm::mlp::mlp_eval_kernel<2,4,1,0,true,false,true>": 64 ** 3,
       "nphm::mlp::mlp_eval_kernel<2,4,1,5,true,false,true>": 64 ** 3, "nphm::mlp::mlp_eval_kernel<4,2,1,5,true,false,true>": 256 ** 3,
       "nphm::mlp::mlp_eval_kernel<1,4,1,0,true,true,false>": 64 ** 3, "nphm::mlp::mlp_eval_kernel<1,4,1,0,true,false,false>": 64 ** 3}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "").replace("void ", "").split("(")[0].replace(", ", ",").replace("(int)", "")
        for k in pts:
            if name.startswith(k.split("<")[0]) and name.replace(" ", "") .startswith(k.replace(" ", "")):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"generated": time.strftime("%Y-%m-%d %H:%M:%S"), "recipe": "tools/pmc_traffic.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB, "
       "FETCH_SIZE x 2 on gfx950, mean over the dispatches of bench.py --steps 2 --warmup 1"}
for k, d in agg.items():
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) if d.get("FETCH_SIZE") else None
    w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) if d.get("WRITE_SIZE") else None
    if f is None or w is None:
        continue
    res[k] = {"fetch_size_kb_per_dispatch": f, "write_size_kb_per_dispatch": w, "traffic_bytes": (2 * f + w) * 1024,
              "points_per_launch": pts[k], "dispatches": len(d["FETCH_SIZE"])}
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
