"""Per-tier phase profile of the identity kernel (verdict round 5, item 1): cycles of every (wavefront, member) visit of a
256^3 launch, split by the wavefront's own tier and by the heaviest tier any wavefront of its workgroup runs the member at.
Needs a timing build and the knobs of the product library (tools/identity_variants.py knobs):
  tools/build_variant.sh prof2 -DNPHM_PROF=2 -DNPHM_DEV_ONLY22      (two stamps per member visit: launch time +2 %)
  tools/build_variant.sh prof1 -DNPHM_PROF=1 -DNPHM_DEV_ONLY22      (+ stamps and a pipeline drain around every chunk: the
                                                                      vmcnt / barrier / GEMM split, at 3x the launch time)
  NPHM_AMD_LIB=$PWD/gpurun_tmp/libprof2.so python tools/tier_profile.py knobs.json [out.json]"""
import json, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _util as U
from nphm_amd import reconstruction as R

MAGIC = 0x7469657270726f66
TIER = ["not needed", "single-term", "two-term", "three-term"]
dev = torch.device("cuda:0")
res = 256
axes = R.grid_axes(U.MINI, U.MAXI, res)
lat = U.sample_latent(0).to(dev)
report = {}
k = json.load(open(sys.argv[1]))
net = U.build_identity(device=dev).eval()
NUM = {"auto": ((k["prune_tol"], k["code"]), None if k["bounds"] is None else torch.tensor(k["bounds"], dtype=torch.float32, device=dev)),
       "f16x3": ((1e-7, net.precision_code("f16x3")), None)}
for mode in ("auto", "f16x3"):
    for it in range(3):
        stats = torch.zeros(128, dtype=torch.int64, device=dev)
        stats[12] = MAGIC
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); R.evaluate_grid(net, lat, axes, stats=stats, numerics=NUM[mode]); t1.record(); torch.cuda.synchronize()
    s = stats.cpu().numpy().astype(float)
    ms = t0.elapsed_time(t1)
    tab = s[16:96].reshape(4, 4, 5)
    total = tab[:, :, 1].sum()
    rows = []
    print(f"== {mode}: launch {ms:.2f} ms (timing build), pairs/point {s[0]/s[1]:.3f}, single {s[15]/s[1]:.3f} two {s[14]/s[1]:.3f} "
          f"three {(s[0]-s[14]-s[15])/s[1]:.3f}; wave-cycles in the member loop {total:.4g}")
    print(f"{'own tier':12s} {'workgroup max':13s} {'visits':>10s} {'share':>6s} {'cyc/visit':>9s} {'gemm+epi':>9s} {'vmcnt':>7s} {'barrier':>8s} {'other':>7s}")
    for t in range(4):
        for T in range(4):
            n, tot, vm, bar, ge = tab[t, T]
            if n == 0: continue
            rows.append(dict(own=TIER[t], wg_max=TIER[T], visits=n, share=tot / total, cycles=tot / n, gemm_epilogue=ge / n,
                             vmcnt_wait=vm / n, barrier_wait=bar / n))
            print(f"{TIER[t]:12s} {TIER[T]:13s} {n:10.0f} {tot/total:6.3f} {tot/n:9.0f} {ge/n:9.0f} {vm/n:7.0f} {bar/n:8.0f} {(tot-ge-vm-bar)/n:7.0f}")
    by_own = {TIER[t]: dict(visits=tab[t, :, 0].sum(), share=tab[t, :, 1].sum() / total,
                            cycles=tab[t, :, 1].sum() / max(tab[t, :, 0].sum(), 1)) for t in range(4)}
    for k, v in by_own.items():
        print(f"  own tier {k:12s}: visits {v['visits']:.0f}, share of loop cycles {v['share']:.3f}, cycles per visit {v['cycles']:.0f}")
    report[mode] = dict(launch_ms_timing_build=ms, pairs_per_point=s[0] / s[1], rows=rows, by_own_tier=by_own,
                        knobs=[NUM[mode][0][0], NUM[mode][0][1]])
if len(sys.argv) > 2:
    json.dump(report, open(sys.argv[2], "w"), indent=1)
