#!/bin/bash
# Run on the GPU box (via gpurun): socket power and shader clock sampled with rocm-smi while (a) the identity lattice kernel,
# (b) the MFMA-only probe, (c) the deformation kernel run back to back for a few seconds each -> gpurun_out/power/*.log
# (evidence for DESIGN 4.1's power-cap argument: what the chip draws and clocks at under each kernel).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/power
mkdir -p "$OUT"
cd "$ROOT"
sample() {   # name, command...
  local name=$1; shift
  "$@" > "$OUT/$name.out" 2>&1 &
  local pid=$!
  sleep 6                                   # import + calibration + warm-up
  : > "$OUT/$name.smi"
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" >> "$OUT/$name.smi"
    echo "--" >> "$OUT/$name.smi"
  done
  wait $pid
}
sample identity python bench.py --workload identity --no-cpu-baseline --no-sub --no-mesh --steps 400 --warmup 5
sample probe python -c "
import ctypes, sys, torch
sys.path.insert(0, '.')
from nphm_amd import _lib
lib = _lib.load()
tf, ghz = ctypes.c_double(), ctypes.c_double()
st = torch.cuda.current_stream(torch.device('cuda:0')).cuda_stream
for _ in range(1400):
    lib.nphm_probe_mfma_rate(ctypes.byref(tf), ctypes.byref(ghz), st)
print(tf.value, ghz.value)
"
sample gemm python -c "
import torch
a = torch.randn(8192, 8192, device='cuda:0', dtype=torch.bfloat16); b = torch.randn(8192, 8192, device='cuda:0', dtype=torch.bfloat16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3000): torch.matmul(a, b)
torch.cuda.synchronize(); e0.record()
for _ in range(6000): torch.matmul(a, b)
e1.record(); torch.cuda.synchronize()
print(2 * 8192 ** 3 * 6000 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 'TFLOP/s')
"
sample two_stage python bench.py --workload two_stage --no-cpu-baseline --steps 60 --warmup 2
python - "$OUT" <<'PY'
import re, sys, glob, os, statistics as st
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*.smi")):
    txt = open(f).read()
    pw = [float(x) for x in re.findall(r"Power \(W\):\s*([0-9.]+)", txt)]
    ck = [float(x) for x in re.findall(r"sclk clock level:?\s*\d*:?\s*\(?([0-9.]+)Mhz", txt)]
    tj = [float(x) for x in re.findall(r"junction\) \(C\):\s*([0-9.]+)", txt)]
    name = os.path.basename(f)[:-4]
    fmt = lambda v: "n/a" if not v else f"median {st.median(v):.0f} (min {min(v):.0f}, max {max(v):.0f}, n {len(v)})"
    print(f"{name:10s} power W: {fmt(pw)}   sclk MHz: {fmt(ck)}   junction C: {fmt(tj)}")
PY
head -12 "$OUT/identity.smi"
