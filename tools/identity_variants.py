"""Same-box timing of library variants of the identity kernel at FIXED knobs (dev tool, GPU box).
  python tools/identity_variants.py knobs gpurun_out/knobs.json         # product library: calibrate `auto`, dump what it runs
  NPHM_AMD_LIB=$PWD/gpurun_tmp/libX.so python tools/identity_variants.py time gpurun_out/knobs.json [modes...]
Modes: auto (the dumped knobs), light / two / heavy (every kept member single- / two- / three-term, prune_tol 1e-7).
A variant built with -DNPHM_DEV_ONLY22 holds eval_kernel<2,2> alone: nothing here launches anything else with it."""
import json, os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _util as U
from nphm_amd import reconstruction as R

dev = torch.device("cuda:0")
res = 256
axes = R.grid_axes(U.MINI, U.MAXI, res)
lat = U.sample_latent(0).to(dev)
net = U.build_identity(device=dev).eval()
cmd, path = sys.argv[1], sys.argv[2]
if cmd == "knobs":
    (tol, code), bounds, _ = net.inference_numerics(dev, lat[None], res ** 3)
    json.dump({"prune_tol": tol, "code": code, "bounds": None if bounds is None else bounds.cpu().numpy().tolist()}, open(path, "w"))
    print("knobs", tol, hex(code))
    sys.exit(0)
k = json.load(open(path))
P = net.precision_code
MODES = {"auto": ((k["prune_tol"], k["code"]), None if k["bounds"] is None else torch.tensor(k["bounds"], dtype=torch.float32, device=dev)),
         "light": ((1e-7, P("f16x3a2", 1.4, 1.4)), None), "two": ((1e-7, P("f16x3a2", 1e-30, 1.4)), None),
         "heavy": ((1e-7, P("f16x3")), None)}
for mode in (sys.argv[3:] or ["auto", "light", "two", "heavy"]):
    num = MODES[mode]
    ts = []
    for it in range(8):
        stats = torch.zeros(128, dtype=torch.int64, device=dev)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); out = R.evaluate_grid(net, lat, axes, stats=None if os.environ.get("NO_STATS") else stats, numerics=num); t1.record(); torch.cuda.synchronize()
        ts.append(t0.elapsed_time(t1))
    s = stats.cpu().numpy().astype(float)
    ms = sorted(ts[2:])[len(ts[2:]) // 2]
    s[1] = max(s[1], 1.0)
    print(f"{mode:6s} {ms:7.2f} ms  {res**3 / ms / 1e3:7.1f} Mpoints/s  pairs/pt {s[0]/s[1]:.3f} (1: {s[15]/s[1]:.3f} 2: {s[14]/s[1]:.3f} 3: {(s[0]-s[14]-s[15])/s[1]:.3f})"
          f"  checksum {float(out.double().sum()):.6f}")
