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
