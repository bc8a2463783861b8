"""An independent geometric check of the CPU checker: closed-form nearest hits in numpy.

The reference holds no vectors for intersection `t` or normals, so the checker (oracle/cpu_ref.cpp) is pinned
only by being a careful restatement. This file is a second restatement that shares nothing with it: whole-array
numpy written from the textbook formulas -- |o + t d - c|^2 = r^2 for spheres (sphere.rs:28-56), the plane
parameter t = (k - o_a) / d_a with the inclusive extent test for rectangles (rect.rs:63-86), the nearest of six
faces for boxes (render_box.rs:117-129) -- evaluated by brute force over all objects in object space
(hittable.rs:204-237: o' = M o, d' = M (o + d) - M o, so t is shared with world space). It is applied to the
committed golden rays of scenes whose objects are spheres, rectangles and boxes, and must give the checker's object
ids and its `t` to 1e-12 relative (both sides are f64 evaluations of the same quantity), normals to 1e-9.

What brute force cannot know is which true hits the reference's bounding-box tree throws away: leaf boxes are the
boxes of only two transformed corners (hittable.rs:239-282), tested per axis (bvh.rs:18-33) against the range that
ends at the nearest hit found so far (bvh.rs:152-159), so on rotated objects (Cube Field) a true hit can be dropped
depending on the order in which the leaves are visited. That rule is restated here too, with the leaf order as
an input (read from the tree the scene builds; inner boxes contain their leaves' boxes and the per-axis test is
monotone in the box, so only leaf boxes can decide).
"""
import os

import numpy as np
import pytest

from ray_tracing_amd.scene import RenderBox, Sphere, XYRect, XZRect, YZRect
from tests.helpers import GOLDEN_DIR, golden_key, scene_by_name

T_MIN = float(np.float32(0.001))             # World::nearest_hit takes f32 bounds (world.rs:112-114)
T_MAX = float(np.finfo(np.float32).max)


def _rect(o, d, axis, a_axis, b_axis, a0, a1, b0, b1, k, normal_sign):
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (k - o[:, axis]) / d[:, axis]
        a = o[:, a_axis] + t * d[:, a_axis]
        b = o[:, b_axis] + t * d[:, b_axis]
        ok = (t >= T_MIN) & (t <= T_MAX) & (a >= a0) & (a <= a1) & (b >= b0) & (b <= b1)
    n = np.zeros((len(t), 3))
    n[:, axis] = normal_sign
    return np.where(ok, t, np.inf), n


def _shape_hit(shape, o, d):
    """(t, object-space normal) per ray; t = inf for a miss."""
    if isinstance(shape, Sphere):
        c, r = np.asarray(shape.origin, dtype=np.float64), float(shape.radius)
        oc = o - c
        a = (d * d).sum(axis=1)
        half_b = (oc * d).sum(axis=1)
        cc = (oc * oc).sum(axis=1) - r * r
        disc = half_b * half_b - a * cc
        with np.errstate(invalid="ignore", divide="ignore"):
            sq = np.sqrt(np.where(disc >= 0.0, disc, np.nan))
            near, far = (-half_b - sq) / a, (-half_b + sq) / a
        near_ok = (near >= T_MIN) & (near <= T_MAX)
        far_ok = (far >= T_MIN) & (far <= T_MAX)
        t = np.where(near_ok, near, np.where(far_ok, far, np.inf))
        t = np.where(disc >= 0.0, t, np.inf)
        p = o + np.where(np.isfinite(t), t, 0.0)[:, None] * d
        return t, (p - c) / r
    if isinstance(shape, XYRect):
        return _rect(o, d, 2, 0, 1, shape.x0, shape.x1, shape.y0, shape.y1, shape.k, -1.0 if shape.flip_normals else 1.0)
    if isinstance(shape, XZRect):
        return _rect(o, d, 1, 0, 2, shape.x0, shape.x1, shape.z0, shape.z1, shape.k, -1.0 if shape.flip_normals else 1.0)
    if isinstance(shape, YZRect):
        return _rect(o, d, 0, 1, 2, shape.y0, shape.y1, shape.z0, shape.z1, shape.k, -1.0 if shape.flip_normals else 1.0)
    if isinstance(shape, RenderBox):
        lo, hi = np.asarray(shape.box_min, dtype=np.float64), np.asarray(shape.box_max, dtype=np.float64)
        best_t = np.full(len(o), np.inf)
        best_n = np.zeros((len(o), 3))
        for axis, (ia, ib) in ((2, (0, 1)), (1, (0, 2)), (0, (1, 2))):        # outward normals on every face
            for k, sign in ((hi[axis], 1.0), (lo[axis], -1.0)):
                t, n = _rect(o, d, axis, ia, ib, lo[ia], hi[ia], lo[ib], hi[ib], k, sign)
                take = t < best_t
                best_t = np.where(take, t, best_t)
                best_n = np.where(take[:, None], n, best_n)
        return best_t, best_n
    raise NotImplementedError(type(shape).__name__)


def _shape_box(shape):
    if isinstance(shape, Sphere):
        c, r = np.asarray(shape.origin, dtype=np.float64), float(shape.radius)
        return c - r, c + r
    if isinstance(shape, XYRect):
        return np.array([shape.x0, shape.y0, shape.k - 0.001]), np.array([shape.x1, shape.y1, shape.k + 0.001])
    if isinstance(shape, XZRect):
        return np.array([shape.x0, shape.k - 0.001, shape.z0]), np.array([shape.x1, shape.k + 0.001, shape.z1])
    if isinstance(shape, YZRect):
        return np.array([shape.k - 0.001, shape.y0, shape.z0]), np.array([shape.k + 0.001, shape.y1, shape.z1])
    return np.asarray(shape.box_min, dtype=np.float64), np.asarray(shape.box_max, dtype=np.float64)


def _two_corner_box_passes(ob, ro, rd, t_far):
    """The leaf box the reference tests: the world-space box of the two transformed corners only, each axis on
    its own against [T_MIN, t_far]."""
    inv_m = np.linalg.inv(ob.transform.world_transform)
    lo, hi = _shape_box(ob.shape)
    a = inv_m[:3, :3] @ lo + inv_m[:3, 3]
    b = inv_m[:3, :3] @ hi + inv_m[:3, 3]
    bmin, bmax = np.minimum(a, b), np.maximum(a, b)
    ok = np.ones(len(ro), dtype=bool)
    with np.errstate(divide="ignore", invalid="ignore"):
        for ax in range(3):
            inv = 1.0 / rd[:, ax]
            t0, t1 = (bmin[ax] - ro[:, ax]) * inv, (bmax[ax] - ro[:, ax]) * inv
            t0, t1 = np.where(inv < 0.0, t1, t0), np.where(inv < 0.0, t0, t1)
            lo_t = np.where(t0 > T_MIN, t0, T_MIN)
            hi_t = np.where(t1 < t_far, t1, t_far)
            ok &= ~(hi_t < lo_t)
    return ok


def leaf_order(info):
    """Object indices in the order the reference's tree visits its leaves (pre-order of the exported tree)."""
    from ray_tracing_amd import World
    b = World(info, device=-1).bvh()
    seen, order = set(), []
    for i in b["object"]:
        if i >= 0 and int(i) not in seen:
            seen.add(int(i))
            order.append(int(i))
    return order


def closed_form_nearest(info, rays):
    ro, rd = rays[:, :3], rays[:, 3:6]
    n_rays = len(rays)
    best_t = np.full(n_rays, np.inf)
    second_t = np.full(n_rays, np.inf)
    best_id = np.full(n_rays, -1, dtype=np.int64)
    best_n = np.zeros((n_rays, 3))
    for i in leaf_order(info):
        ob = info.objects[i]
        m = np.asarray(ob.transform.world_transform, dtype=np.float64)
        o = ro @ m[:3, :3].T + m[:3, 3]
        d = (ro + rd) @ m[:3, :3].T + m[:3, 3] - o
        t, n_obj = _shape_hit(ob.shape, o, d)
        t = np.where(_two_corner_box_passes(ob, ro, rd, np.minimum(best_t, T_MAX)), t, np.inf)
        n_world = n_obj @ np.linalg.inv(m[:3, :3]).T          # hittable.rs:215-220: (M3x3)^-1 n
        take = t < best_t
        second_t = np.where(take, best_t, np.minimum(second_t, t))
        best_n = np.where(take[:, None], n_world, best_n)
        best_id = np.where(take, i, best_id)
        best_t = np.where(take, t, best_t)
    return best_id, best_t, second_t, best_n


@pytest.mark.parametrize("name", ["One Sphere", "Two Sphere", "Easy Cornell Box", "Lambertian Demonstration",
                                  "Metallic Demonstration Smooth", "Light Demonstration", "Cube Field"])
def test_checker_goldens_against_closed_form(name):
    g = np.load(os.path.join(GOLDEN_DIR, golden_key(name) + ".npz"))
    info = scene_by_name(name)
    rays = g["rays"]
    cid, ct, second, cn = closed_form_nearest(info, rays)
    # two surfaces at the same t to rounding: the winner is the tree's visit order, not geometry
    with np.errstate(invalid="ignore"):
        tie = np.isfinite(ct) & (np.abs(second - ct) <= 1e-9 * np.abs(ct))
    gid = g["obj"].astype(np.int64)
    mism = (cid != gid) & ~tie
    assert mism.sum() == 0, f"{int(mism.sum())} of {len(rays)} ids differ from the closed form"
    assert tie.sum() <= 0.01 * len(rays)
    same = (cid == gid) & (gid >= 0)
    assert same.sum() > 0.3 * (gid >= 0).sum() and same.sum() > 200
    np.testing.assert_allclose(g["t"][same], ct[same], rtol=1e-12, atol=0)
    np.testing.assert_allclose(g["normal"][same], cn[same], rtol=0, atol=1e-9)
    pos = rays[same, :3] + ct[same, None] * rays[same, 3:6]
    scale = np.maximum(1.0, np.abs(pos))
    assert (np.abs(g["pos"][same] - pos) <= 1e-9 * scale).all()
    ff = (rays[same, 3:6] * cn[same]).sum(axis=1) <= 0.0
    assert (ff == (g["ff"][same] != 0)).mean() > 0.999          # d . n within rounding of 0 may flip


def test_closed_form_agrees_with_the_live_checker_on_many_rays():
    """Same comparison on 40,000 fresh rays per scene through the checker's nearest-hit probe."""
    from tests import oracle_ffi
    from tests.helpers import probe_rays
    for name in ("Two Sphere", "Easy Cornell Box", "Cube Field"):
        info = scene_by_name(name)
        ow = oracle_ffi.OracleWorld(info)
        rays = probe_rays(ow, info, n_camera=20000, n_random=20000, seed=77)
        oo, to, no, po, fo = ow.trace_rays(rays, seed=1)
        cid, ct, second, cn = closed_form_nearest(info, rays)
        with np.errstate(invalid="ignore"):
            tie = np.isfinite(ct) & (np.abs(second - ct) <= 1e-9 * np.abs(ct))
        assert ((cid != oo) & ~tie).sum() == 0
        same = (cid == oo) & (oo >= 0)
        np.testing.assert_allclose(to[same], ct[same], rtol=1e-12, atol=0)
        np.testing.assert_allclose(no[same], cn[same], rtol=0, atol=1e-9)
