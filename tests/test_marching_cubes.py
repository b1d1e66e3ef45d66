"""CPU tests of the native marching cubes behind mesh_from_logits (the reference delegates this step
to the third-party PyMCubes, utils/reconstruction.py:22-37 — absent here, parity unpinned at the mesh
boundary).  Checked: watertightness / orientation / geometric accuracy on analytic
fields, determinism w.r.t. the thread count, degenerate volumes, and the mesh-level parity measure
used for the north star (Chamfer between meshes of two nearly equal volumes)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import _util as U  # noqa: F401
from nphm_amd import reconstruction as R


def _sphere(res, radius=0.35, centre=(0.03, -0.02, 0.01)):
    ax = np.linspace(-0.5, 0.5, res)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1)
    return (np.linalg.norm(g - np.asarray(centre), axis=-1) - radius).astype(np.float32)


def _edges_manifold(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    directed = e[:, 0].astype(np.int64) * (faces.max() + 1) + e[:, 1]
    rev = e[:, 1].astype(np.int64) * (faces.max() + 1) + e[:, 0]
    # every directed edge appears once and its reverse exactly once (closed, consistently oriented)
    return len(np.unique(directed)) == len(directed) and np.array_equal(np.sort(directed), np.sort(rev))


def chamfer(a, b):
    """Symmetric mean nearest-neighbour distance between vertex sets (the cKDTree form of the
    reference's evaluation/metrics.py:171-194 distance_p2p)."""
    return 0.5 * (cKDTree(b).query(a)[0].mean() + cKDTree(a).query(b)[0].mean())


def test_sphere_is_watertight_oriented_and_accurate():
    res = 48
    sdf = _sphere(res)
    mesh = R.mesh_from_logits(sdf.copy().reshape(-1), [-0.5] * 3, [0.5] * 3, res)
    v, f = np.asarray(mesh.vertices), np.asarray(mesh.faces)
    assert len(v) > 1000 and f.min() >= 0 and f.max() < len(v)
    assert _edges_manifold(f)
    r = np.linalg.norm(v - np.array([0.03, -0.02, 0.01]), axis=1)
    assert np.abs(r - 0.35).max() < 2e-4                      # linear interpolation of a smooth SDF
    # outward normals: signed volume positive and close to the ball's
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol = np.einsum("ij,ij->i", p0, np.cross(p1, p2)).sum() / 6.0
    assert abs(vol - 4 / 3 * np.pi * 0.35 ** 3) < 2e-3


def test_in_place_negation_like_the_reference():
    sdf = _sphere(16).reshape(-1)
    before = sdf.copy()
    R.mesh_from_logits(sdf, [-0.5] * 3, [0.5] * 3, 16)
    assert np.array_equal(sdf, -before)                        # `logits *= -1` on a view


def test_thread_count_does_not_change_the_mesh():
    rng = np.random.default_rng(0)
    vol = rng.standard_normal((20, 17, 23)).astype(np.float32)   # every ambiguous case occurs
    v1, f1 = R.marching_cubes(vol, 0.1, n_threads=1)
    v8, f8 = R.marching_cubes(vol, 0.1, n_threads=8)
    assert np.array_equal(v1, v8) and np.array_equal(f1, f8)
    assert _edges_manifold_open_ok(f1, v1, vol.shape)


def _edges_manifold_open_ok(faces, verts, shape):
    # a random field's surface is open at the volume boundary: interior edges must still pair up
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    key = np.sort(e, axis=1)
    uniq, counts = np.unique(key, axis=0, return_counts=True)
    if counts.max() > 2:
        return False
    single = uniq[counts == 1]
    on_boundary = np.zeros(len(verts), bool)
    for a in range(3):
        on_boundary |= (verts[:, a] == 0) | (verts[:, a] == shape[a] - 1)
    return bool(on_boundary[single].all())


def test_degenerate_volumes():
    v, f = R.marching_cubes(np.ones((4, 4, 4), np.float32), 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f = R.marching_cubes(-np.ones((2, 2, 2), np.float32), 0.0)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    one = -np.ones((3, 3, 3), np.float32); one[1, 1, 1] = 1.0
    v, f = R.marching_cubes(one, 0.0)
    assert len(v) == 6 and len(f) == 8 and _edges_manifold(f)   # octahedron around the single inside voxel
    with pytest.raises(ValueError):
        R.marching_cubes(np.zeros((4, 4), np.float32))


def test_mesh_chamfer_is_the_parity_measure():
    """Two volumes that differ by 1e-6 (the kernels' error level) give meshes whose Chamfer distance
    is far below the north star's 1e-5 bar; a 1e-3 perturbation is detected."""
    res = 40
    sdf = _sphere(res)
    rng = np.random.default_rng(1)
    a = R.mesh_from_logits(sdf.copy().reshape(-1), [-0.5] * 3, [0.5] * 3, res)
    b = R.mesh_from_logits((sdf + 1e-6 * rng.standard_normal(sdf.shape).astype(np.float32)).reshape(-1),
                           [-0.5] * 3, [0.5] * 3, res)
    c = R.mesh_from_logits((sdf + 1e-3).reshape(-1), [-0.5] * 3, [0.5] * 3, res)
    assert chamfer(np.asarray(a.vertices), np.asarray(b.vertices)) < 1e-5
    assert chamfer(np.asarray(a.vertices), np.asarray(c.vertices)) > 1e-4
