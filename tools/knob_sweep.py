"""Dev sweep (GPU; python tools/knob_sweep.py [seeded|trained], NPHM_AMD_PINNED_GUARD=0): identity 256^3 throughput and FULL-volume max error vs the dense three-pass kernel for tier / pruning
knobs around the calibrated ones (is there slack between the rungs of the calibration ladders?)."""
import sys, os, time, argparse, itertools, numpy as np, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
args = argparse.Namespace(res=256, chunk=25000, prune_tol=None)
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "seeded"
ib = bench.IdentityBench(args, dev, 1, 0)
if which == "trained":
    import _util as U
    net, lat = U.build_trained_identity(device=dev)
    ib.net = net.eval(); ib.lat = lat[0].to(dev) if lat.dim() > 1 else lat.to(dev)
net = ib.net
net.numerics = "auto"
net.kernel_knobs(dev, ib.lat[None], 256 ** 3)
c = dict(net.calibration)
print("calibrated:", {k: c[k] for k in ("precision", "light_tol", "mid_tol", "prune_tol", "error")})
orig = net.kernel_knobs
def run(knobs, steps=8):
    net.kernel_knobs = lambda *a, **k: knobs
    for _ in range(2): ib.step("x", True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): ib.step("x", True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    return dt, ib.shard.clone()
_, ref = run((-1.0, net.precision_code("f16x3", None, None, None)), 1)
base = None
L, M_, P = c["light_tol"], c["mid_tol"], c["prune_tol"]
cands = [(L, M_, P)]
for fl, fm, fp in itertools.product((0.71, 1, 1.41, 2), (0.25, 0.5, 1, 2), (1, 2)):
    if (fl, fm, fp) != (1, 1, 1): cands.append((L * fl, M_ * fm, P * fp))
for light, mid, prune in cands:
    code = net.precision_code("f16x3a2", light, mid, c.get("refine_band"))
    dt, vol = run((float(prune), code))
    err = float((vol - ref).abs().max())
    if base is None: base = dt
    print(f"light {light:.2e} mid {mid:.2e} prune {prune:.1e}: {256**3/dt/1e6:7.1f} Mpts/s ({base/dt:5.3f}x)  full-volume err {err:.2e}")
