"""Randomly composed scenes (every shape, material, texture, background and light kind the ABI has, random
transforms, 1-14 objects) through nearest-hit and image parity against the CPU restatement. The
benchmark scenes exercise each kind in one arrangement; this exercises the combinations: span-1 nodes
with stochastic shapes, time-dependent leaf boxes, rotated media, lights of every shape, the tree walk
and the leaf-list walk, and every kernel feature set."""
import numpy as np
import pytest

from ray_tracing_amd import ParallelImage, RayTracer
from ray_tracing_amd import scene as S
from ray_tracing_amd.octree import OctTree, box_points, solid_lambertian, solid_reflect, volume
from ray_tracing_amd.scenarios import synthetic_earthmap
from tests import oracle_ffi
from tests.helpers import probe_rays
from tests.test_gpu_parity import check_ids

pytestmark = pytest.mark.gpu


def random_texture(rng, depth=0):
    k = rng.integers(0, 7 if depth < 2 else 2)
    col = tuple(float(np.float32(c)) for c in rng.uniform(0.05, 0.95, 3))
    if k <= 1:
        return S.SolidColor(col)
    if k == 2:
        return S.CheckerTexture(random_texture(rng, depth + 1), random_texture(rng, depth + 1))
    if k == 3:
        return S.ImageTexture(synthetic_earthmap(64, 32))
    if k == 4:
        return S.MultiplyTexture(random_texture(rng, depth + 1), S.SolidColor(col))
    if k == 5:
        return S.Perlin.default(int(rng.integers(0, 1000)))
    return S.DebugV()


def random_material(rng, emissive=False):
    if emissive:
        e = float(rng.uniform(2.0, 30.0))
        return S.DiffuseLight(S.SolidColor((e, e * 0.9, e * 0.8)))
    k = rng.integers(0, 6)
    if k <= 2:
        return S.Lambertian(random_texture(rng))
    if k == 3:
        return S.Metal(random_texture(rng), float(rng.uniform(0.0, 0.5)))
    if k == 4:
        return S.Dielectric(float(rng.uniform(1.2, 2.0)), tuple(float(np.float32(c)) for c in rng.uniform(0.7, 1.0, 3)))
    return S.Lambertian(S.SolidColor((0.6, 0.6, 0.6)))


def random_solid(rng, material, allow_moving=True):
    k = rng.integers(0, 6 if allow_moving else 5)
    c = rng.uniform(-4.0, 4.0, 3)
    if k == 0:
        return S.Sphere(float(rng.uniform(0.4, 1.5)), tuple(c), material)
    if k == 1:
        a, b = sorted(rng.uniform(-3, 3, 2)), sorted(rng.uniform(-3, 3, 2))
        return S.XYRect(a[0], a[1] + 0.5, b[0], b[1] + 0.5, float(c[2]), material, bool(rng.integers(0, 2)))
    if k == 2:
        a, b = sorted(rng.uniform(-3, 3, 2)), sorted(rng.uniform(-3, 3, 2))
        return S.XZRect(a[0], a[1] + 0.5, b[0], b[1] + 0.5, float(c[1]), material, bool(rng.integers(0, 2)))
    if k == 3:
        a, b = sorted(rng.uniform(-3, 3, 2)), sorted(rng.uniform(-3, 3, 2))
        return S.YZRect(a[0], a[1] + 0.5, b[0], b[1] + 0.5, float(c[0]), material, bool(rng.integers(0, 2)))
    if k == 4:
        lo = c - rng.uniform(0.3, 1.2, 3)
        return S.RenderBox(tuple(lo), tuple(c + rng.uniform(0.3, 1.2, 3)), material)
    c1 = c + rng.uniform(-0.5, 0.5, 3)
    return S.MovingSphere(tuple(c), tuple(c1), 0.0, 1.0, float(rng.uniform(0.3, 1.0)), material)


def random_transform(rng):
    t = S.Transform.identity()
    k = rng.integers(0, 4)
    if k >= 1:
        t = t.translate(tuple(rng.uniform(-1.5, 1.5, 3)))
    if k >= 2:
        t = t.rotate_y(float(rng.uniform(-60, 60)))
    if k == 3:
        t = t.rotate_x(float(rng.uniform(-40, 40)))
    return t


def random_voxels(rng):
    tree = OctTree()
    tree.set_rows(box_points((0, 0, 0), (6, 1, 6)), solid_lambertian((0.7, 0.7, 0.7)))
    rows = [solid_lambertian((0.8, 0.2, 0.2)), solid_reflect((0.8, 0.8, 0.8), 0.1), volume(0.4, (0.3, 0.6, 0.3)),
            volume(0.5, (0.5, 0.5, 0.9), 0.5, solid_lambertian((0.9, 0.9, 0.2)))]
    for _ in range(3):
        lo = rng.integers(0, 4, 3)
        hi = lo + rng.integers(1, 3, 3)
        tree.set_rows(box_points(tuple(lo), tuple(hi)), rows[int(rng.integers(0, len(rows)))])
    return tree.linearise()


def random_scene(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 15))
    objects, light_candidates = [], []
    for i in range(n):
        emissive = rng.random() < 0.25
        kind = rng.random()
        if kind < 0.15 and not emissive:
            shape = S.ConstantMedium(random_solid(rng, S.Lambertian(S.SolidColor((1, 1, 1))), allow_moving=bool(rng.integers(0, 2))),
                                     S.Isotropic(random_texture(rng, 2)), float(rng.uniform(0.05, 1.5)))
        elif kind < 0.25 and not emissive:
            shape = random_voxels(rng)
        else:
            shape = random_solid(rng, random_material(rng, emissive), allow_moving=not emissive)
        ob = S.Object(shape, random_transform(rng))
        objects.append(ob)
        if emissive:
            light_candidates.append(ob)
    lights = [S.Object(o.shape, o.transform) for o in light_candidates[:int(rng.integers(0, 4))]]
    bg_kind = rng.integers(0, 3)
    background = [S.Sky(float(rng.uniform(0.2, 1.0))), S.ConstantColor((0.05, 0.07, 0.1)),
                  S.SunSky(0.4, 0.05, 1.0, 0.9, 20.0)][bg_kind]
    sun = S.Sun(0.9, 1.0, 0.05) if bg_kind == 2 and rng.random() < 0.7 else None
    t1 = float(rng.choice([0.0, 1.0]))
    cam = S.CameraInfo(1.0, 50.0, (0.5, 1.0, 11.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), float(rng.choice([0.0, 0.05])), 11.0, 0.0, t1)
    return S.WorldInfo(objects, lights, background, cam, sun=sun, bvh_seed=int(rng.integers(0, 1 << 30)))


@pytest.mark.parametrize("seed", range(48))
def test_random_scene_parity(seed):
    info = random_scene(1000 + seed)
    ow = oracle_ffi.OracleWorld(info)
    rt = RayTracer(info, device=0, seed=seed)
    a, b = rt.world.bvh(), ow.bvh()
    assert len(a) == len(b) and all((a[k] == b[k]).all() for k in a.dtype.names)
    # nearest hit on camera + interior rays
    rays = probe_rays(ow, info, n_camera=10000, n_random=10000, seed=seed)
    rays[:, 6] = np.random.default_rng(seed).uniform(0.0, max(info.camera.end_time, 1e-9), len(rays))
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=seed)
    oo, to, no, po, fo = ow.trace_rays(rays, seed=seed)
    same = check_ids(og, tg, oo, to)
    hit = (oo >= 0) & same
    np.testing.assert_allclose(tg[hit], to[hit], rtol=1e-5, atol=0)
    np.testing.assert_allclose(ng[hit], no[hit], rtol=0, atol=1e-5)
    assert (fg[hit] == fo[hit]).all()
    # image, sample by sample (the shared addressed random numbers)
    W, H, spp = 48, 36, 4
    rt.world.set_option("trace_zero_throughput", 1)
    img = ParallelImage.new_black(W, H)
    st = rt.trace_part(img.split(1)[0], spp, first_sample=0)
    ref, rst = ow.render(W, H, spp, seed=seed, threads=4)
    tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=2))
    off = np.abs(img.buffer - ref).max(axis=2) > tol
    assert off.mean() < 5e-3, f"{int(off.sum())} of {W * H} pixels differ"
    assert abs(int(st.rays) - int(rst.rays)) <= max(60, rst.rays // 500)
    # both wave-synchronous walks and the per-lane walk agree with each other on the same rays
    # (4: candidate search where the scene qualifies, else the per-lane walk again)
    for mode in (1, 2, 3, 4):
        rt.world.set_option("traversal", mode)
        om, tm, _, _, _ = rt.world.trace_rays(rays, seed=seed)
        assert (om == og).all() and (tm == tg).all()
    rt.world.set_option("traversal", 0)
