"""Volume / reflecting voxels (fast_oct_tree/voxel.rs, ray_trace.rs:118-199): behaviour of the CPU
restatement on small constructed trees (CPU tests), and device == restatement on the same trees
(gpu test). Parity unpinned by the reference: its own tests build solid voxels only."""
import numpy as np
import pytest

from ray_tracing_amd import RayTracer
from ray_tracing_amd.octree import OctTree, box_points, solid_lambertian, solid_reflect, volume
from ray_tracing_amd.scene import CameraInfo, ConstantColor, Object, WorldInfo
from tests.oracle_ffi import OracleWorld


def slab_world(voxel_row, backstop: bool = True, alone: bool = False) -> WorldInfo:
    """An 8 x 8 x 4 slab of `voxel_row` voxels at z in [2, 6) and, behind it, a solid wall at z = 7.
    A second, far-away object keeps the tree from being the only object: a one-object BVH holds the
    object in BOTH leaves of its span-1 root (bvh.rs:224-232), which evaluates a stochastic shape twice
    (`alone=True` builds that case)."""
    from ray_tracing_amd.scene import Lambertian, SolidColor, Sphere
    tree = OctTree()
    tree.set_rows(box_points((0, 0, 2), (8, 8, 6)), voxel_row)
    if backstop:
        tree.set_rows(box_points((0, 0, 7), (8, 8, 8)), solid_lambertian((0.2, 0.3, 0.4)))
    cam = CameraInfo(1.0, 40.0, (4.0, 4.0, -20.0), (4.0, 4.0, 4.0), (0.0, 1.0, 0.0), 0.0, 24.0, 0.0, 0.0)
    objects = [Object(tree.linearise())]
    if not alone:
        objects.append(Object(Sphere(0.5, (500.0, 500.0, 500.0), Lambertian(SolidColor((0.5, 0.5, 0.5))))))
    return WorldInfo(objects, [], ConstantColor((0.0, 0.0, 0.0)), cam)


def rays_into_slab(n, seed):
    rng = np.random.default_rng(seed)
    o = np.stack([rng.uniform(1.0, 7.0, n), rng.uniform(1.0, 7.0, n), np.full(n, -3.0)], 1)
    d = np.stack([rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n), np.ones(n)], 1)
    return np.concatenate([o, d, np.zeros((n, 1))], 1)


def test_dense_volume_stops_rays_just_inside():
    ow = OracleWorld(slab_world(volume(50.0, (0.5, 0.5, 0.5))))
    obj, t, normal, pos, ff = ow.trace_rays(rays_into_slab(2000, 1), seed=3)
    assert (obj == 0).all()
    depth = pos[:, 2] - 2.0
    assert (depth > 0).all() and depth.mean() == pytest.approx(1.0 / 50.0, rel=0.15)   # exponential free path
    assert (normal == np.array([1.0, 0.0, 0.0])).all()                                  # volume hits: unit_x


def test_single_object_tree_is_evaluated_twice():
    """The span-1 root of a one-object BVH holds the octree in both leaves; the second evaluation draws a
    fresh free path and wins when it is shorter: the depth is the minimum of two draws (rate 2 * density)."""
    ow = OracleWorld(slab_world(volume(50.0, (0.5, 0.5, 0.5)), alone=True))
    obj, t, normal, pos, ff = ow.trace_rays(rays_into_slab(4000, 1), seed=3)
    assert (pos[:, 2] - 2.0).mean() == pytest.approx(1.0 / 100.0, rel=0.1)


def test_thin_volume_lets_rays_through_to_the_wall():
    ow = OracleWorld(slab_world(volume(1e-6, (0.5, 0.5, 0.5))))
    obj, t, normal, pos, ff = ow.trace_rays(rays_into_slab(2000, 2), seed=3)
    assert np.allclose(pos[:, 2], 7.0) and (normal == np.array([0.0, 0.0, -1.0])).all()


def test_free_path_distribution_and_escape():
    """density 0.5 over a 4-deep slab: P(stop inside) = 1 - exp(-0.5 * path length); the survivors hit
    the wall (the state is dropped in the empty layer between slab and wall, ray_trace.rs:191-194)."""
    ow = OracleWorld(slab_world(volume(0.5, (0.5, 0.5, 0.5))))
    rays = rays_into_slab(20000, 3)
    obj, t, normal, pos, ff = ow.trace_rays(rays, seed=9)
    inside = pos[:, 2] < 6.0 - 1e-9
    length = 4.0 * np.linalg.norm(rays[:, 3:6], axis=1) / rays[:, 5]
    assert inside.mean() == pytest.approx(np.mean(1.0 - np.exp(-0.5 * length)), abs=0.01)
    assert np.allclose(pos[~inside, 2], 7.0)


def test_edge_effect_probability_and_normal():
    row = volume(1e-6, (0.5, 0.5, 0.5), 0.25, solid_lambertian((0.9, 0.1, 0.1)))
    ow = OracleWorld(slab_world(row))
    obj, t, normal, pos, ff = ow.trace_rays(rays_into_slab(20000, 4), seed=5)
    at_edge = np.isclose(pos[:, 2], 2.0)
    assert at_edge.mean() == pytest.approx(0.25, abs=0.01)
    assert (normal[at_edge] == np.array([0.0, 0.0, -1.0])).all()     # the step's normal, a solid hit
    assert np.allclose(pos[~at_edge, 2], 7.0)


def test_reflecting_voxels_render_finite_and_differ_from_lambertian():
    a = OracleWorld(slab_world(solid_reflect((0.8, 0.8, 0.8), 0.1)))
    b = OracleWorld(slab_world(solid_lambertian((0.8, 0.8, 0.8))))
    ia, _ = a.render(32, 32, 8, seed=1, threads=2)
    ib, _ = b.render(32, 32, 8, seed=1, threads=2)
    assert np.isfinite(ia).all() and np.isfinite(ib).all()
    # black background, no lights: everything is black either way, but the ray counts differ
    # (a mirror sends the ray on, a diffuse surface ends at a zero pdf or the background)
    assert ia.max() == 0.0 and ib.max() == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("row", [volume(50.0, (0.5, 0.5, 0.5)), volume(0.5, (0.5, 0.4, 0.3)),
                                 volume(0.3, (0.5, 0.5, 0.5), 0.25, solid_lambertian((0.9, 0.1, 0.1))),
                                 volume(0.3, (0.5, 0.5, 0.5), 0.6, solid_reflect((0.9, 0.9, 0.1), 0.2)),
                                 solid_reflect((0.8, 0.8, 0.8), 0.1)], ids=["dense", "thin", "edge-lambert", "edge-reflect", "reflect"])
@pytest.mark.parametrize("alone", [False, True], ids=["two-objects", "one-object"])
def test_gpu_matches_restatement_on_slabs(row, alone):
    info = slab_world(row, alone=alone)
    ow = OracleWorld(info)
    rt = RayTracer(info, device=0, seed=2)
    rays = rays_into_slab(50000, 11)
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=4)
    oo, to, no, po, fo = ow.trace_rays(rays, seed=4)
    assert (og == oo).all()
    np.testing.assert_allclose(tg, to, rtol=1e-5, atol=0)
    np.testing.assert_allclose(pg, po, rtol=1e-5, atol=1e-9)
    assert (ng == no).all() and (fg == fo).all()
    from ray_tracing_amd import ParallelImage
    img = ParallelImage.new_black(64, 64)
    rt.world.set_option("trace_zero_throughput", 1)
    st = rt.trace_part(img.split(1)[0], 8, first_sample=0)
    ref, rst = ow.render(64, 64, 8, seed=2, threads=4)
    assert abs(int(st.rays) - int(rst.rays)) <= max(4, rst.rays // 2000)
    np.testing.assert_allclose(img.buffer, ref, rtol=1e-4, atol=1e-5)
