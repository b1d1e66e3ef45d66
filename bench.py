#!/usr/bin/env python3
"""bench.py — SDF-query throughput of the NPHM identity field (BASELINE.json metric).

A "step" = one dense extraction of the NPHM 39-anchor identity SDF on the res^3 lattice of the
reference (bounds fitting_pointclouds.py:166-167): latent prologue (anchors + folded biases) +
fused grid kernel (+ all-gather of the x-slabs when N > 1), output resident in HBM.  Inputs
(packed weights, latent, axis vectors) are resident before the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_DENSE = 9_616_000            # reference formulation, 40 x 2 x 120 200
FLOP_MEMBER_FOLDED = 2 * 81_800   # one member, one point, latent folded
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16x3a": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--prune-tol", type=float, default=None)
    ap.add_argument("--precision", default="bf16x3a", choices=["f32", "bf16x3", "bf16x3a"])
    ap.add_argument("--chunk", type=int, default=25000, help="get_logits chunk whose last voxel is overwritten (eval mode)")
    ap.add_argument("--workload", default="identity", choices=["identity", "two_stage", "npm", "fitting"],
                    help="identity = BASELINE.json configs[1] (the contract line); the others are the remaining "
                         "configs (two_stage = configs[2], npm = configs[0], fitting = configs[4]), single GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-binning", action="store_true", help="brick-order traversal instead of tiles binned by member set")
    ap.add_argument("--no-mesh", action="store_true", help="skip the mesh-extract leg (kernel timing experiments)")
    ap.add_argument("--cpu-sample", type=int, default=40000)
    return ap.parse_args()


def measured_traffic(kernel, n_points):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 PMC passes
    (profiles/traffic.json; the counters cannot be read live from inside the process), scaled to this
    launch's point count.  None if no profile covers the kernel."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
        return t["traffic_bytes"] * n_points / t["points_per_launch"]
    except Exception:
        return None


def cpu_baseline(net, lat, axes, n_sample):
    """The oracle (numpy port of the reference arithmetic) on the host cores, on the first
    n_sample lattice points of the same workload."""
    from oracle import nphm_oracle as O
    import _util as U
    params, amean = U.np_state(net), U.anchors_mean()
    g = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
    # a z-column prefix would be all far-field; take a strided sample of whole-volume points
    idx = np.linspace(0, g.shape[0] - 1, n_sample).astype(np.int64)
    pts = g[idx][None].astype(np.float32)
    latn = lat.cpu().numpy()[None, None]
    threads = 1
    try:                                         # numpy's BLAS pool is what the oracle's GEMMs run on
        from threadpoolctl import threadpool_info
        threads = max([1] + [int(i.get("num_threads", 1)) for i in threadpool_info()])
    except Exception:
        pass
    t0 = time.perf_counter()
    O.nphm_identity_forward(params, amean, pts, latn, training=False)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt / 1e6, "unit": "Mpoints/s", "cores": threads, "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{n_sample} lattice points (uniform stride over the {len(axes[0])}^3 volume), "
                      f"oracle/nphm_oracle.py numpy fp32, dense 40-member evaluation, {dt:.1f} s"}


def _timed(fn, steps, warmup):
    """wall seconds of `steps` calls after `warmup` (synchronised on both sides) + per-call HIP events"""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, [a.elapsed_time(b) for a, b in ev]


def other_workloads(args):
    """BASELINE.json configs other than the contract line, one GPU, same JSON shape."""
    import _util as U
    from nphm_amd import reconstruction as R
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    base = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "data": "synthetic (seeded random-init weights, latents ~ shipped statistics)"}
    if args.workload == "two_stage":
        g = U.golden("deformation")
        inet = U.build_identity(device=dev).eval()
        dnet = U.build_deformation(device=dev).eval()
        lat_id = torch.from_numpy(g["lat"].reshape(-1)[:1344]).to(dev)
        lat_ex = torch.from_numpy(g["lat"].reshape(-1)).to(dev)
        axes = [torch.from_numpy(a).to(dev) for a in R.grid_axes(U.MINI, U.MAXI, args.res)]
        n = args.res ** 3
        anchors = inet.prepare_latent(lat_id[None])[2]
        mlp, cond = R._expr_condition(dnet, lat_ex, anchors, dev)
        dt, _ = _timed(lambda: R.evaluate_grid_two_stage(inet, dnet, lat_id, lat_ex, axes, hack_chunk=args.chunk),
                       args.steps, args.warmup)
        _, k_ms = _timed(lambda: R.evaluate_grid_mlp(mlp, cond, axes, add_input=True), args.steps, 1)
        flops = 3 * 2 * 1_074_688
        ach = flops * n / (np.mean(k_ms) * 1e-3) / 1e12
        out = dict(base, metric="SDF query throughput, deformation -> NPHM identity (two-stage), dense lattice",
                   value=n * args.steps / dt / 1e6, unit="Mpoints/s", ms_per_step=dt / args.steps * 1e3,
                   dtype="bf16x3(split-bf16 MFMA) deformation + " + inet.precision + " identity",
                   config={"workload": f"NPHM identity + forward-deformation field, {args.res}^3 (BASELINE.json configs[2])",
                           "res": args.res},
                   roofline={"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                             "traffic": None, "kernel": "nphm::mlp::mlp_eval_kernel<2,2,1,0> (deformation stage, "
                                                        "the longer of the two kernels)", "kernel_ms": float(np.mean(k_ms)),
                             "executed_flops_per_point": flops}, cpu_baseline=None)
    elif args.workload == "npm":
        from oracle import nphm_oracle as O
        gn = U.golden("npm")
        npm = U.build_npm(device=dev).eval()
        res = 64
        axes = R.grid_axes(U.MINI, U.MAXI, res)
        axes_dev = [torch.from_numpy(a).to(dev) for a in axes]
        lat = torch.from_numpy(gn["lat"][None]).to(dev)
        n = res ** 3
        dt, k_ms = _timed(lambda: R.evaluate_grid_mlp(npm, lat, axes_dev), args.steps, args.warmup)
        flops = 3 * 2 * 6_292_480
        ach = flops * n / (np.mean(k_ms) * 1e-3) / 1e12
        cpu = None
        if not args.no_cpu_baseline:
            pts = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)[:, :20000]
            latn = np.repeat(gn["lat"][None, None], pts.shape[1], axis=1)
            t0 = time.perf_counter()
            O.deepsdf_forward(U.np_state(npm), "", pts, latn, nlayers=8)
            tc = time.perf_counter() - t0
            cpu = {"value": pts.shape[1] / tc / 1e6, "unit": "Mpoints/s", "cores": os.cpu_count(), "kind": "port",
                   "sample": f"first {pts.shape[1]} lattice points, oracle numpy fp32, {tc:.1f} s"}
        out = dict(base, metric="SDF query throughput, NPM global DeepSDF, dense lattice", value=n * args.steps / dt / 1e6,
                   unit="Mpoints/s", ms_per_step=dt / args.steps * 1e3, dtype="bf16x3(split-bf16 MFMA, fp32 accumulate)",
                   config={"workload": "NPM global DeepSDF (lat 512, hidden 1024, 8 layers), 64^3 lattice "
                                       "(BASELINE.json configs[0])", "res": res},
                   roofline={"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                             "traffic": None, "kernel": "nphm::mlp::mlp_eval_kernel<1,4,1,0>", "kernel_ms": float(np.mean(k_ms)),
                             "executed_flops_per_point": flops}, cpu_baseline=cpu)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_fitting as BF
        from nphm_amd import fitting as F
        shape_net = U.build_identity(device=dev)
        expr_net = U.build_deformation(device=dev).eval()
        obs = BF.synthetic_observations(shape_net, dev)
        shape_net.train()
        cfg = {k: dict(v) for k, v in BF.SCHEDULE.items()}
        torch.manual_seed(0)
        F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(BF.LAMBDAS), args.warmup, cfg, verbose=False)
        torch.cuda.synchronize()
        steps = max(args.steps, 20)
        t0 = time.perf_counter()
        F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(BF.LAMBDAS), steps, cfg, verbose=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out = dict(base, steps=steps, metric="latent-code fitting steps/s (inference_iterative_root_finding_joint)",
                   value=steps / dt, unit="steps/s", ms_per_step=dt / steps * 1e3,
                   dtype="bf16x3 kernels + fp32 PyTorch ops",
                   config={"workload": "latent fitting, 3 synthetic observations x 2500 points, 5 x 1000 points per step, "
                                       "Adam on identity + expression codes (BASELINE.json configs[4])"},
                   roofline=None, cpu_baseline=None,
                   note="host-latency bound: ~8 ms of Python/launch overhead per step against ~2 ms of kernels")
    print(json.dumps(out))


def pytorch_rocm_line(net, lat, axes_dev, chunk, n_chunks=20):
    """The same extraction the way the reference runs it on a GPU: chunked get_logits loop over
    PyTorch-ROCm ops (this repo's composite formulation of the module = the reference arithmetic; the
    reference checkout itself is not on the GPU box), on the first n_chunks chunks of the lattice."""
    ax, ay, az = axes_dev
    ry, rz = ay.numel(), az.numel()
    n = n_chunks * chunk
    idx = torch.arange(n, device=ax.device)
    pts = torch.stack([ax[idx // (ry * rz)], ay[(idx // rz) % ry], az[idx % rz]], dim=-1)[None]
    net.backend = "composite"
    try:
        def run():
            out = []
            for p in torch.split(pts, chunk, dim=1):
                with torch.no_grad():
                    sdf, _ = net(p, lat.reshape(1, 1, -1).repeat(1, p.shape[1], 1), None)
                out.append(sdf.squeeze().detach().cpu())
            return torch.cat(out)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        net.backend = "hip"
    return {"value": n / dt / 1e6, "unit": "Mpoints/s",
            "sample": f"first {n_chunks} chunks of {chunk} lattice points, eager PyTorch-ROCm fp32 on the same GPU, "
                      f"per-chunk latent repeat and device->host copy as in get_logits, {dt * 1e3:.0f} ms"}


def main():
    args = parse()
    if args.workload != "identity":
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("--workload other than identity runs on one GPU")
        return other_workloads(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import _util as U
    from nphm_amd import _lib
    from nphm_amd import reconstruction as R

    net = U.build_identity(device=dev).eval()
    if args.prune_tol is not None:
        net.prune_tol = args.prune_tol
    if args.precision is not None:
        net.precision = args.precision
    lat = U.sample_latent(0).to(dev)
    axes = R.grid_axes(U.MINI, U.MAXI, args.res)
    axes_dev = [torch.from_numpy(a).to(dev) for a in axes]
    rx = ry = rz = args.res
    n_total = rx * ry * rz
    plane = ry * rz
    # N > 1: every rank takes the x-planes of every N-th 8-plane brick slab (work-balanced)
    planes = R.cyclic_planes(rx, world, rank)
    planes_dev = torch.from_numpy(planes).to(dev)
    n_planes = len(planes)

    lib = _lib.load()
    stats = torch.zeros(16, dtype=torch.int64, device=dev)
    shard = torch.zeros(max(n_planes, 1) * plane, dtype=torch.float32, device=dev)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    box = {"full": None}
    ws = None if args.no_binning or not n_planes else R.grid_workspace(dev, n_planes, ry, rz)
    ws_ptr, ws_bytes = (None, 0) if ws is None else (ws.data_ptr(), ws.numel())

    def step(timed, ev=None):
        packed, state, _ = net.prepare_latent(lat[None])
        stream = torch.cuda.current_stream(dev).cuda_stream
        if timed:
            ev[0].record()                     # HIP events on the launch stream bracket the dominant kernel
        if n_planes:
            _lib.check(lib.nphm_identity_eval_grid_planes(
                packed.data_ptr(), state.data_ptr(), axes_dev[0].data_ptr(), axes_dev[1].data_ptr(),
                axes_dev[2].data_ptr(), rx, ry, rz, planes_dev.data_ptr(), n_planes, args.chunk,
                float(net.prune_tol), net._precision_code(), shard.data_ptr(),
                stats.data_ptr() if timed else None, ws_ptr, ws_bytes, stream), "eval_grid_planes")
        if timed:
            ev[1].record()
        if world > 1:
            box["full"] = R.gather_planes(shard[: n_planes * plane], rx, plane)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True, events[i])
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- mesh-extract wall-clock (second half of the BASELINE metric): latent -> SDF volume (all
    # ranks) -> host -> marching cubes -> vertices/faces, measured once outside the timed region
    mesh = None
    barrier()
    t_m0 = time.perf_counter()
    if not args.no_mesh:
        step(False)
    barrier()
    t_m1 = time.perf_counter()
    if rank == 0 and not args.no_mesh:
        vol_dev = shard if world == 1 else box["full"]
        vol_host = R.to_host(vol_dev)
        t_m2 = time.perf_counter()
        vh, fh = R.marching_cubes(vol_host.reshape(rx, ry, rz), 0.0, negate=True)     # host extractor
        t_m3 = time.perf_counter()
        m = SimpleNamespace(vertices=vh, faces=fh)
        # the same mesh without the volume leaving the device: GPU marching cubes, only the mesh travels
        torch.cuda.synchronize()
        t_d0 = time.perf_counter()
        vd, fd = R.marching_cubes_device(vol_dev.view(rx, ry, rz), 0.0, negate=True)
        vd_h, fd_h = R.to_host(vd), R.to_host(fd)
        t_d1 = time.perf_counter()
        cold_ms = (t_d1 - t_d0) * 1e3
        # once more, warm (kernels loaded, scratch and pinned staging buffers cached), like the timed kernel steps
        del vd, fd, vd_h, fd_h
        torch.cuda.synchronize()
        t_d0 = time.perf_counter()
        vd, fd = R.marching_cubes_device(vol_dev.view(rx, ry, rz), 0.0, negate=True)
        vd_h, fd_h = R.to_host(vd), R.to_host(fd)
        t_d1 = time.perf_counter()
        mesh = {"wall_ms": (t_m1 - t_m0) * 1e3 + (t_d1 - t_d0) * 1e3, "volume_ms": (t_m1 - t_m0) * 1e3,
                "device_marching_cubes_ms": (t_d1 - t_d0) * 1e3, "device_marching_cubes_first_call_ms": cold_ms,
                "n_vertices": int(len(vd_h)), "n_faces": int(len(fd_h)),
                "reference_order": {"wall_ms": (t_m3 - t_m0) * 1e3, "d2h_ms": (t_m2 - t_m1) * 1e3,
                                    "host_marching_cubes_ms": (t_m3 - t_m2) * 1e3,
                                    "note": "get_logits -> numpy volume on the host -> mesh_from_logits (host marching cubes, <= 16 threads)"},
                "same_mesh": bool(len(vd_h) == len(m.vertices) and np.array_equal(fd_h, np.asarray(m.faces))),
                "note": "wall = latent -> SDF volume (all ranks, all-gathered) -> marching cubes on the GPU -> vertices/faces on the host; "
                        "PyMCubes of the reference is absent, both extractors are this repo's (bit-identical meshes)"}

    if rank == 0:
        n_local = n_planes * plane
        k_ms = float(np.mean(kernel_ms))
        active = stats.cpu().numpy()
        mean_active = float(active[0]) / max(1, args.steps) / n_local     # evaluated member-points / point
        # executed matrix-pipe FLOPs: the split-bf16 path issues 3 bf16 MFMA products per fp32 product
        passes = 1 if net.precision == "f32" else 3
        mean_light = float(active[15]) / max(1, args.steps) / n_local       # single-pass pairs (adaptive mode)
        exec_flops = (passes * (mean_active - mean_light) + mean_light) * FLOP_MEMBER_FOLDED * n_local
        peak = PEAK_TFLOPS[net.precision]
        achieved = exec_flops / (k_ms * 1e-3) / 1e12
        # the events bracket the whole grid call: with binning that is the tile pre-pass + radix sort
        # (together ~1 % of it) + the dominant kernel
        kname = "nphm::eval_kernel<%d,%d>" % (1 if ws is None else 2, min(net._precision_code(), 1))
        out = {
            "metric": "SDF query throughput, NPHM 39-anchor identity field, dense lattice extraction",
            "value": n_total * args.steps / dt / 1e6, "unit": "Mpoints/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "bf16x3(split-bf16 MFMA, fp32 accumulate)",
                      "bf16x3a": "bf16x3 adaptive(split-bf16 MFMA for blend weights >= 1e-3, single-pass bf16 below)"}[net.precision],
            "data": "synthetic (seeded random-init weights, latent ~ shipped mean/std x0.85)",
            "config": {"workload": f"NPHM 39-anchor identity net, {args.res}^3 lattice extraction "
                                   f"(BASELINE.json configs[1]), eval-mode get_logits chunk {args.chunk}",
                       "res": args.res, "prune_tol": net.prune_tol, "precision": net.precision,
                       "parallelism": (f"cyclic 8-plane x-slabs x{world} + all_gather" if world > 1 else "single GPU")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak,
                         "traffic": measured_traffic(kname, n_local),
                         "algorithmic_bytes": 4 * n_local,
                         "kernel": kname, "rank0_planes": n_planes, "binned_tiles": ws is not None,
                         "kernel_ms": k_ms, "points_per_launch": n_local,
                         "executed_flops_per_point": exec_flops / n_local, "mean_single_pass_members": mean_light,
                         "mfma_passes": passes,
                         "mean_active_members": mean_active,
                         "dense_equiv_tflops": FLOP_DENSE * n_local / (k_ms * 1e-3) / 1e12,
                         "note": "achieved counts EXECUTED matrix FLOPs: the adaptive default issues one pass instead "
                                 "of three for ~46% of the evaluated members, so it is faster at a lower FLOP rate "
                                 "(--precision bf16x3: 3 passes everywhere, ~281 Mpoints/s at frac ~0.34); the kernel is "
                                 "held back by the per-chunk weight streaming / barrier and the VALU epilogue threaded through the MFMA chain (ablations in DESIGN.md section 4.1)"},
        }
        out["mesh_extract"] = mesh
        if not args.no_cpu_baseline and world == 1:
            out["pytorch_rocm_baseline"] = pytorch_rocm_line(net, lat, axes_dev, args.chunk)
            out["cpu_baseline"] = cpu_baseline(net, lat, axes, args.cpu_sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
