"""Multi-process (gloo, world_size 2, CPU) tests of the cyclic x-plane sharding + all-gather reassembly used
for multi-GPU lattice extraction.  The per-slab evaluator is injected (a CPU
stand-in with the same contract as evaluate_grid: slab-local values in flattened lattice order), so
what is tested is the partition, the padding of short slabs and the gather order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _util as U  # noqa: F401  (sys.path)
from nphm_amd import reconstruction as R


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rx, ry, rz = shape
        axes = [np.arange(n, dtype=np.float32) for n in shape]
        plane = ry * rz

        def evaluate(planes, out):
            # value = global flat index, as the kernel's plane-list output would hold for f(i) = i
            p = torch.from_numpy(planes.astype(np.int64))
            out.copy_((p[:, None] * plane + torch.arange(plane)[None, :]).reshape(-1).float())

        def evaluate_returning(planes):            # the round-1 callback form: returns the values instead of filling `out`
            p = torch.from_numpy(planes.astype(np.int64))
            return (p[:, None] * plane + torch.arange(plane)[None, :]).reshape(-1).float()

        ok = True
        for unit in (8, 2):
            full = R.evaluate_grid_sharded(None, torch.zeros(1), axes, evaluate=evaluate, unit=unit)
            ok = ok and torch.equal(full, torch.arange(rx * plane, dtype=torch.float32))
        full = R.evaluate_grid_sharded(None, torch.zeros(1), axes, evaluate=evaluate_returning)
        ok = ok and torch.equal(full, torch.arange(rx * plane, dtype=torch.float32))
        # the two-stage sharding cuts a rank's plane set into contiguous runs and evaluates run by run
        calls = []
        orig = R.evaluate_grid_two_stage

        def fake_two_stage(ds, de, es, ee, ax, *, anchors=None, hack_chunk=None, x_range=None, out=None, **kw):
            calls.append(x_range)
            i = torch.arange(x_range[0] * plane, x_range[1] * plane, dtype=torch.float32)
            out.copy_(i)
            return out

        R.evaluate_grid_two_stage = fake_two_stage
        try:
            dec = type("D", (), {"training": True})()
            full = R.evaluate_grid_two_stage_sharded(dec, None, torch.zeros(1), None, axes, unit=2)
        finally:
            R.evaluate_grid_two_stage = orig
        ok = ok and torch.equal(full, torch.arange(rx * plane, dtype=torch.float32))
        ok = ok and all(b - a <= 2 for a, b in calls) and sum(b - a for a, b in calls) == len(R.cyclic_planes(rx, world, rank, 2))
        q.put((rank, bool(ok), R.cyclic_planes(rx, world, rank, 2).tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,world", [((40, 3, 5), 2), ((21, 4, 3), 2), ((5, 4, 3), 2), ((1, 2, 2), 2),
                                         ((512, 2, 3), 8), ((100, 1, 2), 8)])
def test_all_gather_reassembles_the_volume(shape, world):
    """world 2, and the 8-rank layout of BASELINE.json configs[3]: 512 x-planes as cyclic 8-plane slabs (64 planes per
    rank), plus a ragged 100-plane case where ranks hold 16 / 12 / 8 planes and pad their shards"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    planes = dict((r, b) for r, _, b in res)
    assert sorted(sum((planes[r] for r in range(world)), [])) == list(range(shape[0]))


def test_partitions_cover_without_overlap():
    for rx in (1, 5, 8, 256, 257, 512):
        for world in (1, 2, 4, 8):
            spans = [R.slab_bounds(rx, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == rx
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0] and a[0] <= a[1]
            sets = [R.cyclic_planes(rx, world, r) for r in range(world)]
            assert sorted(np.concatenate(sets).tolist()) == list(range(rx))
            assert all(np.all(np.diff(s) > 0) for s in sets if len(s) > 1)
            if rx % (8 * world) == 0:
                assert len({len(s) for s in sets}) == 1


# ---------------------------------------------------------------------------------------------
# numerics of a sharded evaluation: rank 0 decides, the others take its decision (shared_numerics)
# ---------------------------------------------------------------------------------------------
class _StubDecoder:
    """the four members shared_numerics uses; ``decide`` counts the calls of this rank's own decision"""
    numerics = "auto"
    lat_dim = 4

    def __init__(self, rank):
        self.rank, self.decided, self.serial, self.tol = rank, 0, 7, 2e-7

    def _weights_key(self, device):
        return ("w", 0)

    def _latent_digest(self, lat):
        return tuple(lat.reshape(-1).tolist())

    def inference_numerics(self, device, lat, n_points):
        assert self.rank == 0, "only the deciding rank may calibrate"
        self.decided += 1
        bounds = torch.arange(160, dtype=torch.float32).reshape(40, 4) * 0.5 + 1.0
        return (self.tol, 0x2a0b0c06), bounds, self.serial


def _shared_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dec = _StubDecoder(rank)
        sent = [0]
        orig = R._broadcast_from

        def counting(*a, **k):
            sent[0] += 1
            return orig(*a, **k)

        R._broadcast_from = counting
        lat_a, lat_b = torch.zeros(1, 4), torch.ones(1, 4)
        (knobs, bounds), code = R.shared_numerics(dec, lat_a, 1 << 20)
        first = (knobs, bounds.reshape(-1).tolist(), code, sent[0])
        R.shared_numerics(dec, lat_a, 1 << 20)                      # same call: served from the cache, nothing is sent
        after_repeat = sent[0]
        R.shared_numerics(dec, lat_b, 1 << 20)                      # another latent: rank 0 decides again
        R.shared_numerics(dec, lat_a, 1 << 20)                      # the first one is still cached
        after_b = sent[0]
        dec.serial, dec.tol = 8, 1e-7                               # rank 0 re-calibrated while serving latent c:
        lat_c = torch.full((1, 4), 2.0)
        R.shared_numerics(dec, lat_c, 1 << 20)                      # ... the new serial voids every cached decision
        (knobs_a2, _), _ = R.shared_numerics(dec, lat_a, 1 << 20)
        # the dense MLP's tier code rides in the same broadcast; its callable runs on rank 0 only
        mlp = type("M", (), {"precision": "f16x3", "numerics_target": 5e-6, "_lin_params": lambda self: ([], [])})()
        called = [0]

        def code_fn():
            called[0] += 1
            return 0x00f00001

        (_, _), mcode = R.shared_numerics(dec, lat_a, 1 << 21, mlp=mlp, mlp_key=("expr",), mlp_code_fn=code_fn)
        q.put((rank, first, after_repeat, after_b, sent[0], knobs_a2, dec.decided, mcode, called[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_rank_zero_decides_the_numerics_of_a_sharded_evaluation(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = res[0]
    assert r0[1][0] == (2e-7, 0x2a0b0c06) and r0[1][2] is None and r0[1][3] == 1
    assert r0[1][1] == [1.0 + 0.5 * i for i in range(160)]
    assert r0[6] == 5 and r0[8] == 1              # rank 0 decided five times, ran the MLP's callable once
    for r1 in res[1:]:
        # the same decision on every rank, bounds bit for bit; the other ranks never decided anything
        assert r0[1] == r1[1]
        assert r0[2] == r1[2] == 1                # the repeated call sent nothing
        assert r0[3] == r1[3] == 2                # latent b: one more broadcast; latent a again: cached
        assert r0[4] == r1[4] == 5                # latent c, latent a after the new serial, the two-stage call
        assert r0[5] == r1[5] == (1e-7, 0x2a0b0c06)
        assert r1[6] == 0
        assert r0[7] == r1[7] == 0x00f00001 and r1[8] == 0
