"""Which Python line launches which device kernel in one latent-fitting step (development tool, GPU):
eager steps of inference_iterative_root_finding_joint under torch.profiler with stacks, kernels grouped by the
innermost frame inside this repository.

    python tools/fit_step_ops.py [n_profiled_steps]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import _util as U                     # noqa: E402
import bench                          # noqa: E402
import bench_fitting as BF            # noqa: E402
from nphm_amd import fitting as F     # noqa: E402


def main():
    n_prof = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    shape_net = U.build_identity(device=dev)
    expr_net = U.build_deformation(device=dev).eval()
    obs = BF.synthetic_observations(shape_net, dev)
    shape_net.train()
    cfg = lambda: {k: dict(v) for k, v in bench.FIT_SCHEDULE.items()}
    torch.manual_seed(0)
    F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(bench.FIT_LAMBDAS), 8, cfg(), use_graph=False)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()
    skip = ("aten.view", "aten.detach", "aten.expand", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze",
            "aten.transpose", "aten.t.", "aten.alias", "aten._unsafe_view", "aten.reshape", "aten.permute", "aten.unbind",
            "aten.empty", "aten.as_strided", "aten.lift_fresh", "aten.is_pinned", "aten.split", "aten.new_empty")

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not name.startswith(skip):
                site = "autograd / no repo frame"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if fr.filename.startswith(ROOT) and "fit_step_ops" not in fr.filename:
                        site = f"{fr.filename.replace(ROOT + '/', '')}:{fr.lineno} {fr.name}"
                        break
                shapes = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))[:2]
                sites[(name, site, str(shapes))] += 1
            return func(*args, **(kwargs or {}))

    with Spy():
        F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(bench.FIT_LAMBDAS), n_prof, cfg(), use_graph=False)
        torch.cuda.synchronize()
    print(f"{sum(sites.values()) / n_prof:.1f} non-view aten ops per step")
    for (name, site, shapes), n in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print(f"{n / n_prof:5.1f}x {name[:34]:34s} {site[:70]:70s} {shapes[:80]}")


if __name__ == "__main__":
    main()
