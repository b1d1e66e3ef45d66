#!/usr/bin/env python3
"""bench.py — SDF-query throughput of the NPHM identity field (BASELINE.json metric), plus every other
BASELINE config and precision mode as sub-records of the same JSON line.

A "step" = one dense extraction of the NPHM 39-anchor identity SDF on the res^3 lattice of the
reference (bounds fitting_pointclouds.py:166-167): latent prologue (anchors + folded biases) +
fused grid kernel (+ all-gather of the x-slabs when N > 1), output resident in HBM.  Inputs
(packed weights, latent, axis vectors) are resident before the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): the contract fields describe configs[1]
in the default precision; at N = 1 the line also carries
  "precisions"  the same workload with the numerics pinned: bf16x3a2 (the round-2 default), f16x3, bf16x3, f32 (value + roofline each),
  "configs"     configs[0] (NPM 64^3), configs[2] (two-stage 256^3), configs[4] (latent fitting, 250 steps,
                final loss next to the all-composite PyTorch-ROCm loop), 512^3 on one GPU and one training step of the
                identity decoder at nphm.yaml's sizes next to the composite tier,
  "mfma_sustained"  the matrix-pipe rate an MFMA-only loop sustains on THIS box (power-limited clock),
  "cpu_baseline"    the reference's PyTorch operation sequence on the host cores (oracle/torch_reference.py),
  "cpu_baseline_port" the numpy oracle, "pytorch_rocm_baseline" the eager chunk loop on the same GPU.
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
FLOP_DEFORMATION_FOLDED = 2 * 1_074_688   # deformation backbone, latent folded
FLOP_NPM_FOLDED = 2 * 6_292_480
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "bf16x3a": 2500.0, "bf16x3a2": 2500.0, "f16x3": 2500.0, "f16x3a2": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks (bf16 = f16 rate)
DTYPE = {"f32": "f32", "bf16x3": "bf16x3(split-bf16 MFMA, fp32 accumulate)",
         "bf16x3a": "bf16x3 adaptive(split-bf16 MFMA for blend weights >= 1e-3, single-pass bf16 below)",
         "bf16x3a2": "bf16x3 adaptive(split-bf16 MFMA: 3 passes for blend weights >= 1e-2, 2 passes >= 1e-3, single-pass bf16 below)",
         "f16x3": "f16x3(split-f16 MFMA, fp32 accumulate)",
         "f16x3a2": "f16x3 adaptive(split-f16 MFMA: 3 passes for blend weights >= 8e-2, 2 passes >= 8e-3, single-pass f16 below)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--prune-tol", type=float, default=None)
    ap.add_argument("--precision", default="auto", choices=["auto", "f32", "bf16x3", "bf16x3a", "bf16x3a2", "f16x3", "f16x3a2"],
                    help="auto (default) = the module's default: knobs calibrated for the checkpoint (numerics = 'auto')")
    ap.add_argument("--chunk", type=int, default=25000, help="get_logits chunk whose last voxel is overwritten (eval mode)")
    ap.add_argument("--workload", default="all", choices=["all", "identity", "two_stage", "npm", "fitting", "training", "training_corresp"],
                    help="all (default) = the contract line for configs[1] with every other config / precision as "
                         "sub-records; identity = the contract line alone; the others = one of the remaining configs "
                         "as a line of its own (two_stage = configs[2], npm = configs[0], fitting = configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU / eager PyTorch-ROCm baselines")
    ap.add_argument("--no-sub", action="store_true", help="skip the sub-records (kernel timing experiments)")
    ap.add_argument("--no-binning", action="store_true", help="brick-order traversal instead of tiles binned by member set")
    ap.add_argument("--no-mesh", action="store_true", help="skip the mesh-extract leg (kernel timing experiments)")
    ap.add_argument("--cpu-sample", type=int, default=100000, help="lattice points of the PyTorch-CPU baseline (prefix)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU baseline (0 = min(cores, 32): the fastest of 8..256 on the 256-core GPU box)")
    ap.add_argument("--fit-steps", type=int, default=1000, help="n_steps of the fitting config (HIP tier: step_scale 1; composite comparison: 1/4)")
    return ap.parse_args()


def measured_traffic(kernel, n_points):
    """HBM bytes per launch of a kernel, from the COMMITTED rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/pmc_traffic.sh on the GPU box: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 corrections of
    MI355X_MICROARCH.md; hardware counters cannot be read from inside the timed process), scaled to this launch's
    point count.  None if no profile covers the kernel.  `traffic_source` of the roofline object says so."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
        return t["traffic_bytes"] * n_points / t["points_per_launch"]
    except Exception:
        return None


# single-GPU kernel times of the default numerics (profiles/r06_bench.json: ms per launch of the whole lattice, and of
# rank 0's cyclic share of an 8-rank partition of 512^3 measured on one GPU) - inputs of `ranks.expected` only
ONE_GPU_KERNEL_MS = {256: 26.5, 512: 192.5}
RANK0_OF_8_MS_512 = 25.3


def expected_budget(rx, ry, rz, world):
    """Per-step budget of the N-rank lattice evaluation from one-GPU measurements: kernel time of a rank's cyclic share (the
    partition is work-balanced to 1.017 at 8 ranks), the all-gather of the padded shards (each rank RECEIVES (N-1)/N of the
    volume over its 7 xGMI links; 300 GB/s aggregate assumed, point-to-point links ~50 GB/s each way achieved), and the
    reorder pass (one read + one write of the volume at ~4 TB/s).  The all-gather and the reorder of step k run on a side
    stream under the kernel of step k+1: they are exposed only where they exceed it."""
    n = rx * ry * rz
    full = ONE_GPU_KERNEL_MS.get(rx) if rx == ry == rz else None
    kernel = None if full is None else (RANK0_OF_8_MS_512 if (rx == 512 and world == 8) else full / world * 1.02)
    recv = 4.0 * n * (world - 1) / world
    ag = recv / 300e9 * 1e3
    ro = 8.0 * n / 4e12 * 1e3
    return {"kernel_ms_per_rank": None if kernel is None else round(kernel, 2), "allgather_ms": round(ag, 3), "reorder_ms": round(ro, 3),
            "step_ms_if_overlapped": None if kernel is None else round(max(kernel, ag + ro), 2),
            "mpoints_per_s": None if kernel is None else round(n / max(kernel, ag + ro) / 1e3, 0),
            "basis": "one-GPU kernel times (profiles/r06_bench.json), assumed 300 GB/s all-gather ingress; NOT measured on N GPUs"}


TRAFFIC_SOURCE = "profiles/traffic.json (committed rocprofv3 --pmc passes, tools/pmc_traffic.sh), not measured in this run"


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


# ------------------------------------------------------------------------------------------------------
# configs[1]: NPHM identity field on the lattice (the contract line; also the per-precision sub-records)
# ------------------------------------------------------------------------------------------------------
class IdentityBench:
    def __init__(self, args, dev, world, rank, distributed=False):
        import _util as U
        from nphm_amd import _lib
        from nphm_amd import reconstruction as R
        self.args, self.dev, self.world, self.rank = args, dev, world, rank
        self.distributed = distributed       # launched by torch.distributed.run (N = 1 takes the collective path too)
        self.R, self.lib, self._lib = R, _lib.load(), _lib
        self.net = U.build_identity(device=dev).eval()
        self.lat = U.sample_latent(0).to(dev)
        self.axes = R.grid_axes(U.MINI, U.MAXI, args.res)
        self.axes_dev = [torch.from_numpy(a).to(dev) for a in self.axes]
        self.rx = self.ry = self.rz = args.res
        self.plane = self.ry * self.rz
        # N > 1: every rank takes the x-planes of every N-th 8-plane brick slab (work-balanced)
        self.planes = R.cyclic_planes(self.rx, world, rank)
        self.planes_dev = torch.from_numpy(self.planes).to(dev)
        self.n_planes = len(self.planes)
        self.depth = R.shard_depth(self.rx, world)
        # the kernel writes straight into this rank's (padded) shard of the all-gather.  Two shards / gather buffers:
        # the all-gather + reorder of step k run on a side stream while the kernel of step k + 1 fills the other shard
        # (collective overlapped with compute; a shard is reused only after its gather has completed)
        self.shards = [torch.zeros(max(self.depth, 1) * self.plane, dtype=torch.float32, device=dev) for _ in range(2)]
        self.shard = self.shards[0]
        self.gathered = ([torch.empty(world * self.depth * self.plane, dtype=torch.float32, device=dev) for _ in range(2)]
                         if self.distributed else None)
        self.side = torch.cuda.Stream(device=dev) if self.distributed else None
        self.gather_done = [None, None]
        self.k = 0
        self.full = None
        self.collective_events = []      # (start, all-gather done, reorder done) on the side stream, timed steps only
        self.rank_report = None
        self.calibration_ms = 0.0
        self.shared = None               # N > 1: rank 0's ((prune_tol, precision code), member bounds) for this lattice

    def set_precision(self, precision):
        """'auto' = calibrated knobs (the module's default); anything else pins the mode at prune_tol 1e-7 / --prune-tol"""
        net = self.net
        if precision == "auto":
            net.numerics = "auto"
            # calibrate (and verify on this latent) outside the timed region.  N > 1: rank 0 alone does it and every rank runs
            # ITS knobs and member bounds (R.shared_numerics: one broadcast, cached - the shards are slices of one volume);
            # the wall time and a hash of what each rank runs go into `ranks`
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if self.distributed:
                self.shared = self.R.shared_numerics(net, self.lat[None], self.rx * self.ry * self.rz)[0]
            else:
                net.kernel_knobs(self.dev, self.lat[None], self.rx * self.ry * self.rz)
            torch.cuda.synchronize()
            self.calibration_ms = (time.perf_counter() - t0) * 1e3
            if self.distributed:
                return net._MODE_NAMES[self.shared[0][1] & 0xff]
            c = net.calibration
            return c["precision"]
        self.shared = None
        net.precision = precision                              # pins the numerics
        net.light_tol, net.mid_tol, net.refine_band = None, None, None
        net.prune_tol = self.args.prune_tol if self.args.prune_tol is not None else 1e-7
        return precision

    def step(self, precision, binned, stats=None, ev=None):
        net, R = self.net, self.R
        if self.distributed and net.numerics == "auto":      # rank 0's decision (cached: a steady-state step sends nothing)
            self.shared = R.shared_numerics(net, self.lat[None], self.rx * self.ry * self.rz)[0]
        packed, state, _ = R._identity_state(net, self.lat[None], self.rx * self.ry * self.rz,
                                             self.shared if self.distributed else None)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        ws = R.grid_workspace(self.dev, self.n_planes, self.ry, self.rz) if binned and self.n_planes else None
        i = self.k & 1
        self.k += 1
        self.shard = self.shards[i]
        main = torch.cuda.current_stream(self.dev)
        if self.gather_done[i] is not None:
            main.wait_event(self.gather_done[i])          # the gather that read this shard two steps ago has finished
        if ev is not None:
            ev[0].record()                     # HIP events on the launch stream bracket the dominant kernel
        if self.n_planes:
            self._lib.check(self.lib.nphm_identity_eval_grid_planes(
                packed.data_ptr(), state.data_ptr(), self.axes_dev[0].data_ptr(), self.axes_dev[1].data_ptr(),
                self.axes_dev[2].data_ptr(), self.rx, self.ry, self.rz, self.planes_dev.data_ptr(), self.n_planes,
                self.args.chunk, *state.nphm_knobs, self.shard.data_ptr(),
                None if stats is None else stats.data_ptr(), None if ws is None else ws.data_ptr(),
                0 if ws is None else ws.numel(), stream), "eval_grid_planes")
        if ev is not None:
            ev[1].record()
        if self.distributed:
            import torch.distributed as dist
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                cev = None if ev is None else [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                if cev:
                    cev[0].record(self.side)
                dist.all_gather_into_tensor(self.gathered[i], self.shard)
                if cev:
                    cev[1].record(self.side)
                self.full = R.reorder_gathered(self.gathered[i], self.rx, self.plane, self.world)
                if cev:
                    cev[2].record(self.side)
                    self.collective_events.append(cev)
                done = torch.cuda.Event()
                done.record(self.side)
                self.gather_done[i] = done

    def barrier(self):
        if self.distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def measure(self, precision, steps, warmup, binned=True):
        """(seconds of `steps` steps [max over ranks], mean kernel ms, kernel counters)"""
        stats = torch.zeros(16, dtype=torch.int64, device=self.dev)
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for _ in range(warmup):
            self.step(precision, binned)
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(precision, binned, stats, events[i])
        self.barrier()
        dt = time.perf_counter() - t0
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
        self.rank_report = None
        if self.distributed:
            import torch.distributed as dist
            ag = [a.elapsed_time(b) for a, b, _ in self.collective_events]
            ro = [b.elapsed_time(c) for _, b, c in self.collective_events]
            self.collective_events = []
            # per-rank figures of the timed region: what makes an N > 1 line diagnosable (load balance, how much of the
            # all-gather + reorder stays exposed behind the next step's kernel)
            knob_hash = 0.0
            if self.shared is not None:                       # a float64 carries 52 bits of the digest: enough to see a disagreement
                import hashlib
                (tol, code), bnd = self.shared
                blob = repr((float(tol), int(code))).encode() + (b"" if bnd is None else bnd.cpu().numpy().tobytes())
                knob_hash = float(int(hashlib.sha1(blob).hexdigest()[:13], 16))
            mine = torch.tensor([dt, k_ms, float(np.mean(ag)) if ag else 0.0, float(np.mean(ro)) if ro else 0.0,
                                 float(self.n_planes), self.calibration_ms, knob_hash], dtype=torch.float64, device=self.dev)
            allr = [torch.zeros_like(mine) for _ in range(self.world)]
            dist.all_gather(allr, mine)
            allr = torch.stack(allr).cpu().numpy()
            if len(set(allr[:, 6].tolist())) != 1:
                raise RuntimeError("ranks ran different numerics (knob / member-bound digests %r): the gathered volume is not the "
                                   "single-GPU volume" % (allr[:, 6].tolist(),))
            dt = float(allr[:, 0].max())
            self.rank_report = {"step_ms": [round(v / steps * 1e3, 3) for v in allr[:, 0]], "kernel_ms": [round(v, 3) for v in allr[:, 1]],
                                "allgather_ms": [round(v, 3) for v in allr[:, 2]], "reorder_ms": [round(v, 3) for v in allr[:, 3]],
                                "planes": [int(v) for v in allr[:, 4]],
                                # one-off per rank and weight version, outside the timed region; kernel_ms includes the tile
                                # pre-pass and the radix sort of the rank's planes (they run inside the C call)
                                "calibration_ms": [round(v, 1) for v in allr[:, 5]],
                                "knobs_agree": bool(len(set(allr[:, 6].tolist())) == 1),
                                "kernel_max_over_mean": float(allr[:, 1].max() / max(allr[:, 1].mean(), 1e-12)),
                                "exposed_ms_per_step": float(dt / steps * 1e3 - allr[:, 1].max()),
                                # what this line should read if the partition behaves: the budget a first
                                # measured N > 1 record is read against - ONE-GPU measurements, never a measured scaling figure
                                "expected": expected_budget(self.rx, self.ry, self.rz, self.world)}
        return dt, k_ms, stats.cpu().numpy()

    def record(self, precision, steps, warmup, binned=True):
        """value + roofline of one precision mode (the contract line's fields for this rank layout)"""
        requested, precision = precision, self.set_precision(precision)
        dt, k_ms, active = self.measure(precision, steps, warmup, binned)
        n_total = self.rx * self.ry * self.rz
        n_local = max(1, self.n_planes * self.plane)
        mean_active = float(active[0]) / max(1, steps) / n_local            # evaluated member-points / point
        passes = 1 if precision == "f32" else 3     # the split paths issue 3 bf16 / f16 MFMA products per fp32 product
        mean_light = float(active[15]) / max(1, steps) / n_local            # single-pass pairs (adaptive modes)
        mean_mid = float(active[14]) / max(1, steps) / n_local              # two-pass pairs (bf16x3a2)
        exec_flops = (passes * (mean_active - mean_light - mean_mid) + 2 * mean_mid + mean_light) * FLOP_MEMBER_FOLDED * n_local
        peak = PEAK_TFLOPS[precision]
        achieved = exec_flops / (k_ms * 1e-3) / 1e12
        # what the matrix pipe actually ISSUES: 32x32 tiles pad 101 -> 128 / 200 -> 224 rows and 200 -> 208 / 104 -> 112
        # columns - 583 / 391 / 199 MFMAs of 32768 FLOP per (wavefront of 32 points, member) with 3 / 2 / 1 passes
        mfma = {3: 583, 2: 391, 1: 199}
        issued_flops = ((mean_active - mean_light - mean_mid) * mfma[3] + mean_mid * mfma[2] + mean_light * mfma[1]) * 32768 / 32 * n_local \
            if precision != "f32" else None
        # the events bracket the whole grid call: with binning that is the tile pre-pass + radix sort
        # (together ~1 % of it) + the dominant kernel
        kname = "nphm::eval_kernel<%d,%d>" % (2 if binned else 1, 0 if precision == "f32" else 2 if precision.startswith("f16") else 1)
        tkey = kname if precision in ("bf16x3a", "bf16x3a2", "f16x3a2") else kname + ":" + precision
        return {
            "value": n_total * steps / dt / 1e6, "unit": "Mpoints/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "dtype": (DTYPE[precision] if requested != "auto" else
                      {"f16x3a2": "f16x3 adaptive(split-f16 MFMA, fp32 accumulate: 3 / 2 / 1 passes per member by the size of its blend term, "
                                  "thresholds calibrated per checkpoint - config.numerics)",
                       "f16x3": "f16x3(split-f16 MFMA, fp32 accumulate; calibrated: no cheaper tier met the target - config.numerics)"}[precision]),
            "numerics": self.numerics_report(requested),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": measured_traffic(tkey, n_local), "traffic_source": TRAFFIC_SOURCE,
                         "algorithmic_bytes": 4 * n_local, "kernel": kname, "rank0_planes": self.n_planes, "binned_tiles": binned,
                         "kernel_ms": k_ms, "points_per_launch": n_local,
                         "executed_flops_per_point": exec_flops / n_local, "mean_single_pass_members": mean_light,
                         "mean_two_pass_members": mean_mid,
                         "mfma_passes": passes, "mean_active_members": mean_active,
                         "issued_tflops_with_tile_padding": None if issued_flops is None else issued_flops / (k_ms * 1e-3) / 1e12,
                         "dense_equiv_tflops": FLOP_DENSE * n_local / (k_ms * 1e-3) / 1e12},
        }

    def numerics_report(self, requested):
        net = self.net
        if requested != "auto":
            return {"mode": "fixed", "precision": net.precision, "prune_tol": net.prune_tol}
        c = net.calibration
        if c is None:                                  # N > 1, rank != 0: this rank runs rank 0's decision and calibrated nothing
            (tol, code), bnd = self.shared
            return {"mode": "auto (rank 0's calibration, broadcast: reconstruction.shared_numerics)", "prune_tol": tol,
                    "precision_code": code, "member_bounds": bnd is not None, "calibration_ms": round(self.calibration_ms, 1)}
        return {"mode": "auto (calibrated per checkpoint against the dense exact-fp32 kernel)", "precision": c["precision"],
                "light_tol": c["light_tol"], "mid_tol": c["mid_tol"], "prune_tol": c["prune_tol"], "refine_band": c["refine_band"],
                "member_bounds": c["bounds"] is not None,
                "sample_max_abs_err": c["error"], "target": c["target"], "sample_points": c["n_points"],
                "terms_per_sample_point": c.get("terms_per_point"), "settings_searched": len(c.get("searched", ())),
                # one-off per weight version, outside the timed region (bounds fit + search + verification of this latent)
                "calibration_ms": round(self.calibration_ms, 1)}

    def mesh_extract(self, precision, binned):
        """second half of the BASELINE metric: latent -> SDF volume (all ranks) -> marching cubes -> vertices/faces,
        measured once outside the timed region"""
        R, rx, ry, rz = self.R, self.rx, self.ry, self.rz
        self.barrier()
        t_m0 = time.perf_counter()
        self.step(precision, binned)
        self.barrier()
        t_m1 = time.perf_counter()
        if self.rank != 0:
            return None
        vol_dev = self.full if self.distributed else self.shard[: rx * self.plane]
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
        return {"wall_ms": (t_m1 - t_m0) * 1e3 + (t_d1 - t_d0) * 1e3, "volume_ms": (t_m1 - t_m0) * 1e3,
                "device_marching_cubes_ms": (t_d1 - t_d0) * 1e3, "device_marching_cubes_first_call_ms": cold_ms,
                "n_vertices": int(len(vd_h)), "n_faces": int(len(fd_h)),
                # reference order: get_logits -> numpy volume on the host -> mesh_from_logits (host marching cubes)
                "reference_order": {"wall_ms": (t_m3 - t_m0) * 1e3, "d2h_ms": (t_m2 - t_m1) * 1e3,
                                    "host_marching_cubes_ms": (t_m3 - t_m2) * 1e3},
                "same_mesh": bool(len(vd_h) == len(m.vertices) and np.array_equal(fd_h, np.asarray(m.faces)))}


# ------------------------------------------------------------------------------------------------------
# configs[2], configs[0], configs[4]
# ------------------------------------------------------------------------------------------------------
def _mlp_exec_flops(mlp, mask, single_mask=0):
    """executed MFMA FLOPs per point of the dense skip-MLP kernel: hidden GEMM layer l runs 1 term of the split product if bit
    l of ``single_mask`` is set, 2 if bit l of ``mask`` is (DeepSDF's calibrated tiers), else 3; lin0's coordinate step counts
    3; the last linear layer (out_dim rows) runs in fp32 on the VALU since round 5 and counts as one pass of its 2 out k
    FLOPs.  Also returns the per-layer pass list."""
    d_in = mlp.lat_dim + mlp.input_dim
    total, passes = 0.0, []
    for l in range(mlp.num_layers - 1):
        W = getattr(mlp, f"lin{l}").weight
        out_f, in_f = W.shape
        k = 3 if l == 0 else (in_f - d_in if l in mlp.skip_in else in_f) + (3 if l in mlp.skip_in else 0)
        hidden = 0 < l < mlp.num_layers - 2
        p = 3 if l == 0 else 1 if (not hidden or (single_mask >> l) & 1) else 2 if (mask >> l) & 1 else 3
        passes.append(p)
        total += p * 2.0 * out_f * k
    return total, passes


def _mlp_kernel_name(mlp, shape, num):
    """symbol of the lattice launch (csrc/mlp_kernel.hip launch_eval): format and, when every hidden GEMM layer runs the two-term
    product, the variant that keeps four K-steps of wh in flight"""
    f16 = mlp.precision == "f16x3"
    hidden = num["passes_per_layer"][1:-1]
    if not f16:
        return f"nphm::mlp::mlp_eval_kernel<{shape},1,0,false>"
    if hidden and all(p == 1 for p in hidden):
        shape = {"2,2": "4,2", "1,4": "2,4"}[shape]          # single-term everywhere: twice the points per workgroup
        return f"nphm::mlp::mlp_eval_kernel<{shape},1,0,true,false,true>"
    if len(hidden) >= 2 and all(p == 1 for p in hidden[:-1]) and hidden[-1] == 2:
        shape = {"2,2": "4,2", "1,4": "2,4"}[shape]          # ... but the last hidden layer, two-term: in two K halves through the
        kind = 6 if getattr(mlp, "tail_k_split", False) else 5   # workspace (KIND 6, the default) or in two point halves (KIND 5)
        return f"nphm::mlp::mlp_eval_kernel<{shape},1,{kind},true,false,true>"
    no_wl = bool(hidden) and all(p <= 2 for p in hidden)     # no three-term layer: the variant without wl registers
    return f"nphm::mlp::mlp_eval_kernel<{shape},1,0,true,{'true' if no_wl else 'false'},false>"


def _mlp_numerics_report(mlp):
    r = dict(mlp.last_numerics or {})
    mask = int(r.get("mask", 0))
    single_mask = int(r.get("single_mask", 0))
    flops, passes = _mlp_exec_flops(mlp, mask, single_mask)
    return {"precision": mlp.precision, "numerics": mlp.numerics, "single_mask": single_mask, "two_pass_mask": mask,
            "passes_per_layer": passes, "points_per_workgroup": (2 if (r.get("single_term") or r.get("tail_two_term")) else 1) * (64 if mlp.hidden_dim <= 512 else 32),
            "target": r.get("target"), "sample_err": r.get("err"), "verified_err": r.get("verified_err"),
            "all_single_err": r.get("all_single_err")}, flops


def two_stage_record(args, dev, steps, warmup):
    """configs[2]: deformation -> identity on the 256^3 lattice (get_logits_backward semantics with anchors)"""
    import _util as U
    from nphm_amd import reconstruction as R
    g = U.golden("deformation")
    inet = U.build_identity(device=dev).eval()
    dnet = U.build_deformation(device=dev).eval()
    lat_id = torch.from_numpy(g["lat"].reshape(-1)[:1344]).to(dev)
    lat_ex = torch.from_numpy(g["lat"].reshape(-1)).to(dev)
    axes = [torch.from_numpy(a).to(dev) for a in R.grid_axes(U.MINI, U.MAXI, args.res)]
    n = args.res ** 3
    anchors = inet.prepare_latent(lat_id[None])[2]
    mlp, cond = R._expr_condition(dnet, lat_ex, anchors, dev)
    dt, _ = _timed(lambda: R.evaluate_grid_two_stage(inet, dnet, lat_id, lat_ex, axes, hack_chunk=args.chunk), steps, warmup)
    _, k_ms = _timed(lambda: R.evaluate_grid_mlp(mlp, cond, axes, add_input=True), steps, 1)
    num, flops = _mlp_numerics_report(mlp)
    ach = flops * n / (np.mean(k_ms) * 1e-3) / 1e12
    kname = _mlp_kernel_name(mlp, "2,2", num)
    return {"metric": "SDF query throughput, deformation -> NPHM identity (two-stage), dense lattice",
            "value": n * steps / dt / 1e6, "unit": "Mpoints/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "dtype": f"{mlp.precision} (split-f16 MFMA, fp32 accumulate; {num['passes_per_layer']} product terms per layer, calibrated) deformation + "
                     + (inet.calibration or {}).get("precision", inet.precision) + " identity",
            "numerics": num,
            "config": {"workload": f"NPHM identity + forward-deformation field, {args.res}^3 (BASELINE.json configs[2])",
                       "res": args.res},
            "roofline": {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                         "traffic": measured_traffic(kname, n), "algorithmic_bytes": 12 * n,
                         "kernel": kname + " (deformation stage, the longer of the two kernels)",
                         "kernel_ms": float(np.mean(k_ms)), "executed_flops_per_point": flops}}


def npm_record(args, dev, steps, warmup, cpu):
    """configs[0]: NPM global DeepSDF on the 64^3 lattice; its CPU side = the reference's PyTorch-CPU path"""
    import _util as U
    from nphm_amd import reconstruction as R
    gn = U.golden("npm")
    npm = U.build_npm(device=dev).eval()
    res = 64
    axes = R.grid_axes(U.MINI, U.MAXI, res)
    axes_dev = [torch.from_numpy(a).to(dev) for a in axes]
    lat = torch.from_numpy(gn["lat"][None]).to(dev)
    n = res ** 3
    dt, k_ms = _timed(lambda: R.evaluate_grid_mlp(npm, lat, axes_dev), steps, warmup)
    num, flops = _mlp_numerics_report(npm)
    ach = flops * n / (np.mean(k_ms) * 1e-3) / 1e12
    kname = _mlp_kernel_name(npm, "1,4", num)
    out = {"metric": "SDF query throughput, NPM global DeepSDF, dense lattice", "value": n * steps / dt / 1e6,
           "unit": "Mpoints/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "dtype": f"{npm.precision} (split-f16 MFMA, fp32 accumulate; {num['passes_per_layer']} product terms per layer, calibrated)",
           "numerics": num,
           "config": {"workload": "NPM global DeepSDF (lat 512, hidden 1024, 8 layers), 64^3 lattice "
                                  "(BASELINE.json configs[0])", "res": res},
           "roofline": {"bound": "mfma", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0,
                        "traffic": measured_traffic(kname, n), "algorithmic_bytes": 4 * n, "kernel": kname,
                        "kernel_ms": float(np.mean(k_ms)), "executed_flops_per_point": flops}}
    # the same lattice with the tier target relaxed to 1e-5 (opt-in: npm.numerics_target; still a decade inside the 1e-4 bar):
    # every hidden layer single-term -> the 64-points-per-workgroup variant, half the weight bytes per point.  Reported BESIDE
    # the default, never instead of it
    keep_target, keep_cache = npm.numerics_target, npm._two_pass_cache
    try:
        npm.numerics_target, npm._two_pass_cache = 1e-5, None
        dt2, k2 = _timed(lambda: R.evaluate_grid_mlp(npm, lat, axes_dev), steps, warmup)
        num2, flops2 = _mlp_numerics_report(npm)
        out["relaxed_target_1e-5"] = {"value": n * steps / dt2 / 1e6, "unit": "Mpoints/s", "kernel_ms": float(np.mean(k2)), "numerics": num2,
                                      "frac": flops2 * n / (np.mean(k2) * 1e-3) / 1e12 / 2500.0, "kernel": _mlp_kernel_name(npm, "1,4", num2)}
    finally:
        npm.numerics_target, npm._two_pass_cache = keep_target, keep_cache
    if cpu:
        # the whole config on the host cores, the way the reference runs it: chunked get_logits, fp32 PyTorch-CPU - the
        # reference's own DeepSDF + get_logits when oracle/_ref holds their bytecode (oracle/build_ref.py), else the restatement
        from oracle import ref_loader as RL
        sd = {k: v.detach().cpu() for k, v in npm.state_dict().items()}
        grid = torch.from_numpy(R.create_grid_points_from_bounds(U.MINI, U.MAXI, res)).float()[None]
        enc = torch.from_numpy(gn["lat"])[None, None]
        if RL.available():
            kind, ref_net, ref_gl = "reference", RL.build_npm(sd).eval(), RL.load("reconstruction").get_logits
            run = lambda p: ref_gl(ref_net, enc, p, args.chunk)
            what = "the reference's own DeepSDF + get_logits (oracle/_ref bytecode)"
        else:
            from oracle import torch_reference as T
            kind = "port"
            fwd = lambda p, l: T.deepsdf_forward(sd, "", p, l, nlayers=8)
            run = lambda p: T.get_logits(fwd, enc, p, args.chunk)
            what = "the reference's PyTorch op sequence restated (oracle/torch_reference.py)"
        threads, sweep = _thread_sweep(args, lambda: run(grid[:, :args.chunk]))
        run(grid[:, :args.chunk])                                              # warm-up: one chunk
        t0 = time.perf_counter()
        run(grid)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / tc / 1e6, "unit": "Mpoints/s", "cores": threads, "host_cores": os.cpu_count(),
                               "kind": kind, "thread_sweep_s_per_chunk": sweep,
                               "sample": f"all {n} lattice points, chunked get_logits, {what}, PyTorch-CPU fp32, {threads} threads "
                                         f"(fastest of the sweep), one run after a one-chunk warm-up, {tc:.1f} s"}
    return out


FIT_LAUNCHES_PER_STEP = 26      # kernel launches of one replayed step (profiles/r06_fitting_launches.txt, tools/fit_launches.py; refreshed per round)
FIT_LAMBDAS = {"surface": 2.0, "reg_expr": 0.01, "reg_global": 0.25, "reg_unobserved": 10, "reg_loc": 0.05,
               "symm_dist": 5.0}                                                   # fitting_pointclouds.py:253-259
FIT_SCHEDULE = {"lr": {200: 2, 400: 2, 600: 2, 800: 2}, "symm_dist": {200: 10, 500: 9999},
                "reg_glob": {200: 3, 600: 10}, "reg_loc": {500: 3, 600: 10}, "reg_expr": {600: 10}}   # :261-266


def grid512_record(args, dev, steps=2):
    """configs[3] on ONE GPU: the whole 512^3 lattice in one launch, and rank 0's share of the 8-rank cyclic
    partition (the launch every rank of the 8-GPU job runs; the all-gather of 64 MiB per rank comes on top)."""
    big = argparse.Namespace(**vars(args))
    big.res = 512
    ib = IdentityBench(big, dev, 1, 0)
    full = ib.record(args.precision, steps, 1)
    share = IdentityBench(big, dev, 8, 0)                     # rank 0's plane set of 8, no process group
    dt, k_ms, _ = share.measure(args.precision, steps, 1)
    n_share = share.n_planes * share.plane
    return {"metric": "SDF query throughput, NPHM identity field, 512^3 lattice", "value": full["value"], "unit": "Mpoints/s",
            "ms_per_step": full["ms_per_step"], "steps": steps, "dtype": full["dtype"],
            "config": {"workload": "NPHM 39-anchor identity net, 512^3 lattice on one GPU (BASELINE.json configs[3] needs 8 GPUs: "
                                   "its per-rank launch is timed below)", "res": 512},
            "roofline": full["roofline"],
            "rank0_of_8": {"planes": share.n_planes, "points": n_share, "kernel_ms": k_ms,
                           "projected_8gpu_mpoints_per_s_without_allgather": 512 ** 3 / (k_ms * 1e-3) / 1e6}}


def trained_record(args, dev, steps=3):
    """configs[1] on the TRAINED-LIKE checkpoint (tests/golden/trained_state.npz: 5 000 steps on analytic head-like
    surfaces, weights up to 1.25) with one of its trained codes: throughput and member statistics where the blend
    fields and the SDF have trained sharpness, next to its max |error| against the dense exact-fp32 kernel on a
    64^3 sub-lattice."""
    import _util as U
    from nphm_amd import reconstruction as R
    ib = IdentityBench(args, dev, 1, 0)
    ib.net, codes = U.build_trained_identity(device=dev)
    ib.net.eval()
    ib.lat = codes[0]
    rec = ib.record(args.precision, steps, 1)
    axes = R.grid_axes(U.MINI, U.MAXI, 64)
    fast = R.evaluate_grid(ib.net, ib.lat, axes, hack_chunk=0)
    ib.net.precision, ib.net.prune_tol = "f32", -1.0
    exact = R.evaluate_grid(ib.net, ib.lat, axes, hack_chunk=0)
    r = rec["roofline"]
    return {"metric": "SDF query throughput, NPHM identity field, trained-like checkpoint", "value": rec["value"], "unit": "Mpoints/s",
            "ms_per_step": rec["ms_per_step"], "steps": steps, "dtype": rec["dtype"],
            "config": {"workload": f"NPHM identity net, trained-like checkpoint (tests/golden/trained_state.npz), code 0, {args.res}^3", "res": args.res},
            "numerics": rec["numerics"],
            "roofline": {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms", "mean_active_members",
                                           "mean_single_pass_members", "mean_two_pass_members", "executed_flops_per_point")},
            "max_abs_err_vs_dense_f32_64cubed": float((fast - exact).abs().max()), "max_abs_sdf": float(exact.abs().max())}


def fitting_record(args, dev, with_reference_loop=True):
    """configs[4]: latent fitting at its stated horizon (fitting_pointclouds.py:269-276: step_scale 1, n_steps 1000,
    the published schedule) on the HIP tier; synthetic observations on the level set of a seeded ground-truth
    identity.  Comparison: the SAME loop on the all-composite PyTorch-ROCm tier (the reference's arithmetic on this
    GPU) at step_scale 1/4 (250 steps through the same schedule transitions; 1000 composite steps would take 13 s).
    Bound of the record: the share of a step the GPU is busy replaying the captured graph (HIP events around the
    replay), next to the launch count per step of the committed kernel trace."""
    import _util as U
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_fitting as BF
    from nphm_amd import fitting as F

    def run(backend, step_scale):
        shape_net = U.build_identity(device=dev)
        expr_net = U.build_deformation(device=dev).eval()
        obs = BF.synthetic_observations(shape_net, dev)
        shape_net.train()
        if backend:
            shape_net.backend = backend
            expr_net.backend = backend
        cfg = lambda: {k: dict(v) for k, v in FIT_SCHEDULE.items()}
        torch.manual_seed(0)
        F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(FIT_LAMBDAS), 8, cfg(), verbose=False)   # warm-up
        torch.cuda.synchronize()
        hist, timing = [], {}
        torch.manual_seed(0)
        t0 = time.perf_counter()
        F.inference_iterative_root_finding_joint(shape_net, expr_net, obs, dict(FIT_LAMBDAS), args.fit_steps, cfg(),
                                                 step_scale=step_scale, verbose=False, history=hist, timing=timing)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tail = hist[-20:]
        fc = getattr(expr_net.defDeepSDF, "_fit_cache", None)
        fit_num = {"fit_numerics": expr_net.defDeepSDF.fit_numerics, "two_pass_mask": None if fc is None else int(fc[1]),
                   "sample_err_in_units_of_targets": None if fc is None else fc[2].get("err"),
                   "value_target": expr_net.defDeepSDF.fit_target, "jacobian_target": expr_net.defDeepSDF.fit_jacobian_target,
                   "reverified_err": None if fc is None else fc[2].get("reverified_err"), "verify_every": expr_net.defDeepSDF.fit_verify_every}
        return {"expr_decoder_numerics": fit_num, "steps_per_s": len(hist) / dt, "ms_per_step": dt / len(hist) * 1e3, "steps": len(hist), "step_scale": step_scale,
                "first_surface_loss": hist[0]["surface"], "final_surface_loss": float(np.mean([h["surface"] for h in tail])),
                "final_total_loss": float(np.mean([h["loss"] for h in tail])),
                "final_valid_correspondences": float(np.mean([h["n_valid"] for h in tail])),
                "graph_ms": timing.get("graph_ms"), "graph_steps": timing.get("graph_steps")}

    ours = run(None, 1.0)
    busy = None if not ours["graph_ms"] else ours["graph_ms"] / ours["ms_per_step"]
    out = {"metric": "latent-code fitting steps/s (inference_iterative_root_finding_joint)", "value": ours["steps_per_s"],
           "unit": "steps/s", "ms_per_step": ours["ms_per_step"], "steps": ours["steps"],
           "dtype": "split-f16 (expression decoder: calibrated two-term layers, expr_decoder_numerics) / split-bf16 (identity tier, backward) kernels + fp32 PyTorch ops",
           "expr_decoder_numerics": ours["expr_decoder_numerics"],
           "config": {"workload": "latent fitting, 3 synthetic observations x 2500 points, 5 x 1000 points per step, Adam on "
                                  f"identity + expression codes, n_steps {args.fit_steps} at step_scale 1 = {ours['steps']} steps, "
                                  "the published schedule (BASELINE.json configs[4])"},
           "first_surface_loss": ours["first_surface_loss"], "final_surface_loss": ours["final_surface_loss"],
           "final_total_loss": ours["final_total_loss"], "final_valid_correspondences": ours["final_valid_correspondences"],
           # surface losses = mean over the last 20 steps
           "roofline": {"bound": "latency (launch-bound GPU stream)", "graph_replay_ms": ours["graph_ms"],
                        "gpu_busy_frac_of_step": busy, "graph_steps_timed": ours["graph_steps"],
                        "host_ms_per_step_outside_graph": None if busy is None else ours["ms_per_step"] - ours["graph_ms"],
                        "launches_per_step": FIT_LAUNCHES_PER_STEP, "launches_source": "profiles (rocprofv3 --kernel-trace of --workload fitting)"}}
    if with_reference_loop:
        # the comparison of the END of the fit is made at ONE horizon: both tiers at step_scale 1/4 (250 steps, same seed, same
        # schedule transitions).  (Round 4 divided the 1000-step HIP loss by the 250-step composite loss: not a parity figure.)
        ref = run("composite", 0.25)
        ours_250 = run(None, 0.25)
        out["reference_loop_same_gpu"] = ref          # every field on the composite tier = eager PyTorch-ROCm fp32, 250 steps
        out["hip_loop_same_horizon"] = {k: ours_250[k] for k in ("steps", "step_scale", "final_surface_loss", "final_total_loss",
                                                                "final_valid_correspondences", "steps_per_s")}
        out["final_surface_loss_ratio_same_horizon"] = ours_250["final_surface_loss"] / max(ref["final_surface_loss"], 1e-30)
        out["speedup_vs_reference_loop_same_gpu"] = ours_250["steps_per_s"] / ref["steps_per_s"]
    return out


def training_record(args, dev, with_composite=True, steps=8):
    """one training step of the identity decoder (training.py:110-135: compute_loss, backward w.r.t.
    every weight and the latent codes, gradient clipping, AdamW) at nphm.yaml's sizes - batch 32, 750 face + 50
    non-face + 800 near-surface + 93 far points per subject - on the HIP training tier next to the SAME step on the
    composite PyTorch tier (the reference's arithmetic on this GPU).  Synthetic points, seeded random-init weights."""
    import _util as U
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train as BT
    B, n_face = 32, 750
    batch = BT.synthetic_batch(B, n_face, dev)
    lat0 = torch.stack([U.sample_latent(10 + b) for b in range(B)])[:, None, :].to(dev)
    n_pts = sum(batch[k].shape[1] for k in ("points_face", "points_non_face", "sup_grad_near", "sup_grad_far"))

    def run(backend, operands=None):
        net = U.build_identity(device=dev).train()
        net.train_backend = backend
        if operands is not None:
            net.train_operands = operands
        lat = lat0.clone().requires_grad_()
        opt = torch.optim.AdamW(list(net.parameters()) + [lat], lr=5e-4, weight_decay=0.01)
        torch.cuda.reset_peak_memory_stats()
        losses = [float(BT.step(net, lat, batch, opt)) for _ in range(2)]
        torch.cuda.synchronize()
        net._train_backward_events = events = []          # HIP events around the reverse + weight-gradient kernels
        t0 = time.perf_counter()
        losses += [BT.step(net, lat, batch, opt) for _ in range(steps)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        net._train_backward_events = None
        hbm = None
        if events:
            ms = float(np.mean([a.elapsed_time(b) for a, b, _ in events]))
            nbytes = float(np.mean([n for _, _, n in events]))
            hbm = {"bound": "hbm", "achieved": nbytes / ms / 1e6, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / ms / 1e6 / 8000.0,
                   # reverse + weight-gradient kernels of one step (HIP events on the launch stream); achieved = the stored
                   # operands of the weight gradients written once and read once / the time of the two kernels
                   "traffic": nbytes, "kernel_ms": ms}
        return {"ms_per_step": dt * 1e3, "roofline": hbm, "steps_per_s": 1.0 / dt, "first_loss": losses[0], "last_loss": float(losses[-1]),
                "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "prune_tol": net.prune_tol,
                # "auto" (the default) resolves by batch size (FastEnsembleDeepSDFMirrored.TRAIN_F16_MIN_POINTS)
                "operands": (("f16" if B * n_pts >= net.TRAIN_F16_MIN_POINTS else "f32") if net.train_operands == "auto" else net.train_operands)}

    ours = run("hip")
    out = {"metric": "identity-decoder training steps/s (compute_loss + backward + AdamW)", "value": ours["steps_per_s"],
           "unit": "steps/s", "ms_per_step": ours["ms_per_step"], "steps": steps,
           "dtype": f"bf16x3 kernels (member MLPs, their double backward); weight-gradient operands stored as {ours['operands']} "
                    "(f16: binary16 with per-stream scales, one-pass contraction; f32: split three-pass) + fp32 PyTorch ops (blend, losses, optimizer)",
           "config": {"workload": f"training step, batch {B} x {n_pts} points (nphm.yaml: 750 face + 50 non-face + 800 near + 93 far), "
                                  "loss terms of loss_functions.py:20-110 with create_graph gradients, all decoder weights + latent codes trainable "
                                  "(SURVEY 8 f4)", "prune_tol": ours["prune_tol"]},
           "first_loss": ours["first_loss"], "last_loss": ours["last_loss"], "peak_mem_gb": ours["peak_mem_gb"],
           "roofline": ours["roofline"]}
    for other in ("f32", "f16", "bf16"):                   # the other storage formats of the weight gradients' operands (opt-in)
        if other != ours["operands"]:
            o = run("hip", other)
            out["operands_" + other] = {"ms_per_step": o["ms_per_step"], "roofline": o["roofline"], "steps_per_s": o["steps_per_s"],
                                        "last_loss": o["last_loss"], "peak_mem_gb": o["peak_mem_gb"]}
    if with_composite:
        ref = run("composite")
        ref.pop("roofline", None)
        out["composite_same_gpu"] = ref               # the same step on the composite PyTorch tier (fp32 autograd double backward)
        out["speedup_vs_composite"] = ref["ms_per_step"] / ours["ms_per_step"]
        out["last_loss_diff"] = abs(ours["last_loss"] - ref["last_loss"])
    return out


def training_corresp_record(args, dev, steps=8):
    """SURVEY 8 f4, widened: one step of the reference's SECOND training stage (training_corresp.py: compute_loss_corresp_forward,
    loss_functions.py:282-326, backward w.r.t. every weight of the deformation network, both code tables and the identity decoder's
    anchor head, Adam) at nphm_def.yaml's sizes - batch 32 x 1000 correspondences + 100 free samples per subject - with the
    dense backbone on the training kernels (csrc/dense_train_kernels.hip) next to the SAME step on the composite PyTorch tier
    (nn.Linear / Softplus under autograd: library GEMMs).  Synthetic points, seeded random-init weights."""
    import _util as U
    from nphm_amd.loss_functions import compute_loss_corresp_forward
    B, n = 32, 1000
    g = torch.Generator().manual_seed(17)
    neutral = (torch.rand(B, n, 3, generator=g) - 0.5) * torch.tensor([0.5, 0.6, 0.5])
    batch = {"points_neutral": neutral, "points_posed": torch.cat([neutral + 0.01 * torch.randn(B, n, 3, generator=g),
                                                                    torch.randn(B, n, 3, generator=g)], -1),
             "gt_anchors": torch.from_numpy(U.anchors_mean()).reshape(1, 39, 3).repeat(B, 1, 1),
             "subj_ind": torch.arange(B).reshape(B, 1) % 24, "idx": torch.arange(B).reshape(B, 1)}
    batch = {k: v.to(dev) for k, v in batch.items()}
    lam = {"corresp": 100.0, "lat_reg": 0.01, "loss_reg_zero": 5.0}

    def run(backend):
        torch.manual_seed(3)
        shape_net = U.build_identity(device=dev).train()
        expr_net = U.build_deformation(device=dev).train()
        expr_net.defDeepSDF.train_backend = backend
        lat_shape, lat_expr = torch.nn.Embedding(24, 1344).to(dev), torch.nn.Embedding(B, 200).to(dev)
        with torch.no_grad():
            lat_shape.weight.copy_(torch.stack([U.sample_latent(40 + i) for i in range(24)]).to(dev))
            lat_expr.weight.mul_(0.1)
        params = list(expr_net.parameters()) + list(shape_net.mlp_pos.parameters()) + list(lat_shape.parameters()) + list(lat_expr.parameters())
        opt = torch.optim.Adam(params, lr=5e-4)

        def step():
            opt.zero_grad(set_to_none=True)
            losses = compute_loss_corresp_forward(batch, expr_net, shape_net, lat_expr, lat_shape, dev, epoch=3)
            total = sum(lam[k] * losses[k] for k in lam)
            total.backward()
            opt.step()
            return total.detach()

        losses = [float(step()) for _ in range(3)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = [step() for _ in range(steps)][-1]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return {"ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "first_loss": losses[0], "last_loss": float(last)}

    ours, ref = run("hip"), run("composite")
    pts = B * (n + 100)
    flops = 3 * 2 * pts * (235 * 512 + 512 * 512 + 512 * 277 + 512 * 512 + 2 * 512 * 512 + 512 * 3)     # forward + two backward products per layer
    return {"metric": "deformation-network training steps/s (compute_loss_corresp_forward + backward + Adam)", "value": ours["steps_per_s"],
            "unit": "steps/s", "ms_per_step": ours["ms_per_step"], "steps": steps,
            "dtype": "split-bf16 x3 MFMA (fp32 accumulate) for the six hidden layers and their data / weight gradients; fp32 PyTorch ops for the "
                     "concatenations, the 3-row output layer, the compressor, losses and the optimizer",
            "config": {"workload": f"train_corresp step, batch {B} x ({n} correspondences + 100 free samples), nphm_def.yaml sizes and lambdas, "
                                   "'compress' deformation network (6 x 512), every weight + both code tables trainable (SURVEY 8 f4)"},
            "first_loss": ours["first_loss"], "last_loss": ours["last_loss"],
            "model_tflops_per_step": flops / 1e12, "model_tflops_per_s": flops / 1e12 / (ours["ms_per_step"] * 1e-3),
            "composite_same_gpu": ref, "speedup_vs_composite": ref["ms_per_step"] / ours["ms_per_step"],
            "last_loss_diff": abs(ours["last_loss"] - ref["last_loss"])}


# ------------------------------------------------------------------------------------------------------
# baselines
# ------------------------------------------------------------------------------------------------------
def _cpu_threads(args):
    return args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 32)


def _thread_sweep(args, run_chunk):
    """PyTorch-CPU thread count for the baseline: --cpu-threads if given, else the fastest of 32 / 64 / 128 / all host cores
    on one chunk each (after one warm-up chunk per setting).  -> (threads, {threads: seconds per chunk})"""
    host = os.cpu_count() or 1
    if args.cpu_threads > 0:
        torch.set_num_threads(args.cpu_threads)
        return args.cpu_threads, {}
    cands = sorted({min(t, host) for t in (32, 64, 128, host)})
    secs = {}
    for t in cands:
        torch.set_num_threads(t)
        run_chunk()
        t0 = time.perf_counter()
        run_chunk()
        secs[t] = time.perf_counter() - t0
    best = min(secs, key=secs.get)
    torch.set_num_threads(best)
    return best, {str(k): round(v, 3) for k, v in secs.items()}


def cpu_baseline(net, lat, axes, args):
    """The reference's PyTorch-CPU path on the host cores, chunked by ITS get_logits, on the first n_sample lattice points of
    the same workload (dense 40-member evaluation: the cost does not depend on where the points lie).  kind "reference":
    the reference's own FastEnsembleDeepSDFMirrored + get_logits, imported from the bytecode oracle/build_ref.py compiled
    from /root/reference in the build container (oracle/_ref/: git-ignored, travels with the snapshot); kind "port" (when
    that bytecode is absent): oracle/torch_reference.py, the same operation sequence restated (bit-identical outputs,
    tests/test_torch_reference.py).  Threads: the fastest of 32 / 64 / 128 / all cores on one chunk; then one warm-up
    chunk + 3 runs of the sample, median."""
    import _util as U
    from oracle import ref_loader as RL
    n = args.cpu_sample
    idx = np.arange(n)
    ry, rz = len(axes[1]), len(axes[2])
    pts = torch.from_numpy(np.stack([axes[0][idx // (ry * rz)], axes[1][(idx // rz) % ry], axes[2][idx % rz]], -1))[None]
    enc = lat.detach().cpu().reshape(1, 1, -1)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    if RL.available():
        kind = "reference"
        ref_net = RL.build_identity(torch.from_numpy(U.anchors_mean()).float().reshape(1, 1, -1, 3), sd).eval()
        ref_get_logits = RL.load("reconstruction").get_logits
        run = lambda p: ref_get_logits(ref_net, enc, p, args.chunk)
        what = ("the reference's own modules (src/NPHM/models/EnsembledDeepSDF.py + reconstruction.get_logits, bytecode under "
                "oracle/_ref built by oracle/build_ref.py)")
    else:
        from oracle import torch_reference as T
        kind = "port"
        amean = torch.from_numpy(U.anchors_mean()).float()
        fwd = lambda p, l: T.nphm_identity_forward(sd, amean, p, l, training=False)[0]
        run = lambda p: T.get_logits(fwd, enc, p, args.chunk)
        what = "the reference's PyTorch op sequence restated (oracle/torch_reference.py; oracle/_ref is absent)"
    threads, sweep = _thread_sweep(args, lambda: run(pts[:, :args.chunk]))
    run(pts[:, :args.chunk])
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        run(pts)
        runs.append(time.perf_counter() - t0)
    dt = float(np.median(runs))
    return {"value": n / dt / 1e6, "unit": "Mpoints/s", "cores": threads, "host_cores": os.cpu_count(), "kind": kind,
            "thread_sweep_s_per_chunk": sweep,
            "sample": f"first {n} lattice points of the {len(axes[0])}^3 volume in chunks of {args.chunk}, {what}, PyTorch-CPU fp32, "
                      f"dense 40-member evaluation, {threads} threads (fastest of the sweep), 1 warm-up chunk + 3 runs, median {dt:.1f} s (runs: "
                      + ", ".join(f"{r:.1f}" for r in runs) + ")"}


def cpu_baseline_port(net, lat, axes, n_sample=20000):
    """The numpy oracle on the host cores (second field; slower than the PyTorch sequence above)."""
    from oracle import nphm_oracle as O
    import _util as U
    params, amean = U.np_state(net), U.anchors_mean()
    g = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
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
    return {"value": n_sample / dt / 1e6, "unit": "Mpoints/s", "cores": threads, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{n_sample} lattice points (uniform stride over the volume), oracle/nphm_oracle.py numpy fp32, "
                      f"dense 40-member evaluation, {dt:.1f} s"}


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


def mfma_sustained(dev):
    """Matrix-pipe rate of an MFMA-only loop on this box (nphm_probe_mfma_rate: every SIMD issues dependent-free
    v_mfma_f32_32x32x16_bf16 with non-trivial operands for ~5 ms): what the chip's power management lets the matrix
    cores sustain - the practical ceiling under the 2.5 PFLOP/s datasheet peak that `roofline.peak` uses."""
    import ctypes
    from nphm_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, "nphm_probe_mfma_rate"):
        return None
    tf, ghz = ctypes.c_double(), ctypes.c_double()
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):
        _lib.check(lib.nphm_probe_mfma_rate(ctypes.byref(tf), ctypes.byref(ghz), stream), "nphm_probe_mfma_rate")
    out = {"tflops": tf.value, "clock_ghz": ghz.value, "frac_of_peak": tf.value / 2500.0}
    # an independent reference on the same box: the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt) at a size that
    # is MFMA-bound, on normally distributed operands - the rate a tuned dense kernel reaches under the same power limit
    try:
        n = 8192
        a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        for _ in range(100):              # (a 10-launch burst reads 1.16 PFLOP/s where 6 000 launches sustain 1.41: let it settle)
            torch.matmul(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 300
        for _ in range(reps):
            torch.matmul(a, b)
        e1.record()
        torch.cuda.synchronize()
        out["library_gemm_bf16_tflops"] = 2.0 * n ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12
        out["library_gemm_shape"] = f"torch.matmul bf16 {n}^3 randn, {reps} launches after 100 warm-up launches"
    except Exception as e:            # noqa: BLE001 - a reference figure only
        out["library_gemm_bf16_tflops"] = None
        out["library_gemm_shape"] = repr(e)
    return out


# ------------------------------------------------------------------------------------------------------
def single_workload(args):
    """--workload two_stage | npm | fitting | training: one of the other configs as a line of its own (one GPU)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    base = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "data": "synthetic (seeded random-init weights, latents ~ shipped statistics)"}
    if args.workload == "two_stage":
        rec = two_stage_record(args, dev, args.steps, args.warmup)
    elif args.workload == "npm":
        rec = npm_record(args, dev, args.steps, args.warmup, not args.no_cpu_baseline)
    elif args.workload == "training":
        rec = training_record(args, dev, with_composite=not args.no_cpu_baseline, steps=max(args.steps, 5))
    elif args.workload == "training_corresp":
        rec = training_corresp_record(args, dev, steps=max(args.steps, 5))
    else:
        rec = fitting_record(args, dev, with_reference_loop=not args.no_cpu_baseline)
    rec.setdefault("cpu_baseline", None)
    print(json.dumps(dict(base, **rec)))


def summary_of(out):
    """Headline numbers of every sub-record in < 1 500 characters, printed as the LAST key of the line (a driver that
    keeps only the tail of a long line still sees mesh_extract and configs[0] / [2] / [4])."""
    r3 = lambda v: None if v is None else float(f"{v:.4g}")
    s = {"value": r3(out["value"]), "frac": r3(out["roofline"]["frac"]), "kernel_ms": r3(out["roofline"]["kernel_ms"])}
    m = out.get("mesh_extract")
    if m:
        s["mesh_extract_ms"] = {"wall": r3(m["wall_ms"]), "volume": r3(m["volume_ms"]), "gpu_mc": r3(m["device_marching_cubes_ms"]),
                                "reference_order_wall": r3(m["reference_order"]["wall_ms"]), "same_mesh": m["same_mesh"]}
    if out.get("mfma_sustained"):
        s["mfma_sustained_tflops"] = r3(out["mfma_sustained"]["tflops"])
        s["library_gemm_bf16_tflops"] = r3(out["mfma_sustained"].get("library_gemm_bf16_tflops"))
    for k, v in (out.get("precisions") or {}).items():
        s["prec_" + k] = [r3(v["value"]), r3(v["roofline"]["frac"])]
    c = out.get("configs") or {}
    if "npm_64" in c:
        s["cfg0_npm_64"] = {"mpts": r3(c["npm_64"]["value"]), "frac": r3(c["npm_64"]["roofline"]["frac"]),
                            "cpu_mpts": r3((c["npm_64"].get("cpu_baseline") or {}).get("value"))}
    if "two_stage_256" in c:
        s["cfg2_two_stage_256"] = {"mpts": r3(c["two_stage_256"]["value"]), "frac": r3(c["two_stage_256"]["roofline"]["frac"])}
    if "grid512_one_gpu" in c:
        s["cfg3_512_one_gpu"] = {"mpts": r3(c["grid512_one_gpu"]["value"]), "rank0_of_8_ms": r3(c["grid512_one_gpu"]["rank0_of_8"]["kernel_ms"])}
    if "fitting" in c:
        f = c["fitting"]
        s["cfg4_fitting"] = {"steps_per_s": r3(f["value"]), "steps": f["steps"], "final_surface_loss": r3(f["final_surface_loss"]),
                             "gpu_busy": r3(f["roofline"]["gpu_busy_frac_of_step"]),
                             "composite_steps_per_s": r3((f.get("reference_loop_same_gpu") or {}).get("steps_per_s")),
                             "loss_ratio_250_steps": r3(f.get("final_surface_loss_ratio_same_horizon"))}
    if "training" in c:
        t = c["training"]
        s["f4_training"] = {"steps_per_s": r3(t["value"]), "ms": r3(t["ms_per_step"]),
                            "composite_ms": r3((t.get("composite_same_gpu") or {}).get("ms_per_step")),
                            "hbm_frac": r3((t.get("roofline") or {}).get("frac"))}
    if "training_corresp" in c:
        t = c["training_corresp"]
        s["f4_training_corresp"] = {"ms": r3(t["ms_per_step"]), "composite_ms": r3(t["composite_same_gpu"]["ms_per_step"]),
                                    "model_tflops_per_s": r3(t["model_tflops_per_s"])}
    if "trained_checkpoint_256" in c:
        t = c["trained_checkpoint_256"]
        s["trained_ckpt"] = {"mpts": r3(t["value"]), "frac": r3(t["roofline"]["frac"]), "members": r3(t["roofline"]["mean_active_members"]),
                             "max_err": r3(t["max_abs_err_vs_dense_f32_64cubed"])}
    if out.get("cpu_baseline"):
        s["cpu_mpts"] = [r3(out["cpu_baseline"]["value"]), out["cpu_baseline"]["cores"]]
    if out.get("pytorch_rocm_baseline"):
        s["pytorch_rocm_mpts"] = r3(out["pytorch_rocm_baseline"]["value"])
    return s


def two_stage_sharded(args, world):
    """--workload two_stage under torch.distributed.run: configs[2] sharded like configs[3] (cyclic 8-plane slabs,
    deformation + identity kernels per slab, one all-gather + reorder), max over ranks, one line on rank 0."""
    import torch.distributed as dist
    import _util as U
    from nphm_amd import reconstruction as R
    rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = int(os.environ.get("NPHM_BENCH_DEVICE", local_rank))
    backend = os.environ.get("NPHM_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    g = U.golden("deformation")
    inet, dnet = U.build_identity(device=dev).eval(), U.build_deformation(device=dev).eval()
    lat_id = torch.from_numpy(g["lat"].reshape(-1)[:1344]).to(dev)
    lat_ex = torch.from_numpy(g["lat"].reshape(-1)).to(dev)
    axes = [torch.from_numpy(a).to(dev) for a in R.grid_axes(U.MINI, U.MAXI, args.res)]
    fn = lambda: R.evaluate_grid_two_stage_sharded(inet, dnet, lat_id, lat_ex, axes, hack_chunk=args.chunk)
    for _ in range(args.warmup):
        fn()
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    dist.barrier(); torch.cuda.synchronize()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t)
    allt = [float(x.item()) for x in allt]
    if rank == 0:
        n, dt = args.res ** 3, max(allt)
        mlp = dnet.defDeepSDF
        num, _ = _mlp_numerics_report(mlp)          # (rank 0 decided the tiers: reconstruction.shared_numerics)
        print(json.dumps({"metric": "SDF query throughput, deformation -> NPHM identity (two-stage), dense lattice", "value": n * args.steps / dt / 1e6,
                          "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": f"{mlp.precision} (split-f16 MFMA, fp32 accumulate; {num['passes_per_layer']} product terms per layer, calibrated) deformation + "
                     + (inet.calibration or {}).get("precision", inet.precision) + " identity",
            "numerics": num, "data": "synthetic (seeded random-init weights)",
                          "config": {"workload": f"NPHM identity + forward-deformation field, {args.res}^3 (BASELINE.json configs[2]), sharded",
                                     "res": args.res, "parallelism": f"cyclic 8-plane x-slabs x{world} + all_gather"},
                          "ranks": {"step_ms": [round(x / args.steps * 1e3, 3) for x in allt]}, "roofline": None, "cpu_baseline": None}))
    dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.workload == "two_stage" and world > 1:
        return two_stage_sharded(args, world)
    if args.workload not in ("all", "identity"):
        if world != 1:
            raise SystemExit("--workload other than all / identity / two_stage runs on one GPU")
        return single_workload(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # dry runs of the N > 1 code path on a box with ONE GPU: NPHM_BENCH_DEVICE=0 puts every rank on that device and
    # NPHM_BENCH_DIST_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one device); timings are meaningless
    dev_index = int(os.environ.get("NPHM_BENCH_DEVICE", local_rank))
    backend = os.environ.get("NPHM_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # launched by torch.distributed.run (any N)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    ib = IdentityBench(args, dev, world, rank, distributed)
    binned = not args.no_binning
    rec = ib.record(args.precision, args.steps, args.warmup, binned)
    mesh = None if args.no_mesh else ib.mesh_extract(args.precision, binned)

    if rank == 0:
        out = {
            "metric": "SDF query throughput, NPHM 39-anchor identity field, dense lattice extraction",
            "value": rec["value"], "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": rec["dtype"], "data": "synthetic (seeded random-init weights, latent ~ shipped mean/std x0.85)",
            "config": {"workload": f"NPHM 39-anchor identity net, {args.res}^3 lattice extraction "
                                   f"(BASELINE.json configs[1]), eval-mode get_logits chunk {args.chunk}",
                       "res": args.res, "numerics": rec["numerics"], "precision": args.precision,
                       "parallelism": (f"cyclic 8-plane x-slabs x{world} + all_gather (of step k, on a side stream, under the "
                                       "kernel of step k+1)" if distributed else "single GPU")},
            # achieved counts EXECUTED matrix FLOPs (tile padding excluded); peak is the datasheet figure - an MFMA-only loop
            # sustains `mfma_sustained.tflops` on this box (power-limited clock), DESIGN.md 4.1
            "roofline": rec["roofline"],
            "mesh_extract": mesh,
        }
        if ib.rank_report is not None:
            out["ranks"] = ib.rank_report
        if world == 1 and args.workload == "all" and not args.no_sub:
            sub_steps = max(2, min(args.steps, 5))
            out["mfma_sustained"] = mfma_sustained(dev)
            out["precisions"] = {p: ib.record(p, sub_steps if p != "f32" else 2, 1, binned)
                                 for p in ("bf16x3a2", "f16x3", "bf16x3", "f32") if p != args.precision}
            out["configs"] = {
                "npm_64": npm_record(args, dev, max(sub_steps, 5), 2, not args.no_cpu_baseline),
                "two_stage_256": two_stage_record(args, dev, sub_steps, 1),
                "fitting": fitting_record(args, dev, with_reference_loop=not args.no_cpu_baseline),
                "grid512_one_gpu": grid512_record(args, dev),
                "training": training_record(args, dev, with_composite=not args.no_cpu_baseline),
                "training_corresp": training_corresp_record(args, dev),
                "trained_checkpoint_256": trained_record(args, dev),
            }
            ib.set_precision(args.precision)
        if not args.no_cpu_baseline and world == 1:
            out["pytorch_rocm_baseline"] = pytorch_rocm_line(ib.net, ib.lat, ib.axes_dev, args.chunk)
            out["cpu_baseline"] = cpu_baseline(ib.net, ib.lat, ib.axes, args)
            out["cpu_baseline_port"] = cpu_baseline_port(ib.net, ib.lat, ib.axes)
        else:
            out["cpu_baseline"] = None
        out["summary"] = summary_of(out)          # LAST key: the sub-records' headline numbers survive a truncated tail
        print(json.dumps(out))
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    # ONE JSON line on stdout: whatever the modules print while they are built (the mirrored constructors repeat the reference's
    # "creating DeepSDF with ..." lines) goes to stderr; only json.dumps above writes to the real stdout
    import contextlib
    _real_stdout = sys.stdout
    _json_print = print

    def print(*a, **k):                                   # noqa: A001 - the three json.dumps prints of this module
        k.setdefault("file", _real_stdout)
        _json_print(*a, **k)
        _real_stdout.flush()

    with contextlib.redirect_stdout(sys.stderr):
        main()
