"""Dev tool (GPU): the search trace of calibrate_numerics on the seeded and the trained-like checkpoint (setting, sample error, cost)."""
import sys, os, torch
ROOT = os.getcwd(); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util as U
from nphm_amd import numerics as NU
dev = torch.device("cuda:0")
for which in ("seeded", "trained"):
    if which == "seeded":
        net = U.build_identity(device=dev).eval(); lat = U.sample_latent(0).to(dev)[None]
    else:
        net, codes = U.build_trained_identity(device=dev); net.eval(); lat = codes[0][None]
    c = NU.calibrate_numerics(net, lat, device=dev)
    print(which, "->", {k: c[k] for k in ("precision", "light_tol", "mid_tol", "prune_tol", "error")})
    print("   terms per point", c["terms_per_point"])
    for setting, e, cost in c["searched"]:
        print(f"   {setting['precision']:8s} light {setting['light_tol']} mid {setting['mid_tol']} prune {setting['prune_tol']}: {e:.2e}  cost {cost:.3f}")
    # tiers without pruning, pruning without tiers at the chosen knobs
    import math
