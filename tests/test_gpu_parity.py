"""Parity of the HIP path (through the C ABI) with the CPU checker and the committed goldens.

Tier A: nearest-hit queries on fixed rays -- object ids bit-exact, t / normals within 1e-5 relative
        (north-star tolerance).
Tier P: path level -- both sides address the same counter-based random numbers, so whole images
        agree to f32 rounding except for rare discrete flips caused by 1-ulp differences in
        transcendental functions / FMA contraction.
Tier B: statistical -- RMSE(GPU, CPU independent estimator) <= 1.15 * RMSE(CPU, CPU') at equal spp,
        channel means within 0.5 % (+ noise allowance).
Full-size properties: union of sample ranges, strip independence, energy / mean consistency.
"""
import ctypes as C
import os

import numpy as np
import pytest

from ray_tracing_amd import ParallelImage, RayTracer, World, scenarios
from tests import oracle_ffi
from tests.helpers import golden_key, probe_rays, rmse, scene_by_name

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["Cornell Smoke", "Random Scene", "Light Demonstration", "Easy Cornell Box", "Two Sphere",
          "Fast Oct Tree Sphere", "Voxel City", "Easy Scene", "Fast Oct Tree 2 density",
          "Fast Oct Tree Volume Lambertian", "Fast Oct Tree Volume Metal", "Fast Oct Tree Volume Ice",
          "Lambertian Demonstration", "Metallic Demonstration Smooth", "Metallic Demonstration Rough",
          "Dielectric Demonstration, Low Refraction", "Dielectric Demonstration, High Refraction", "Cube Field",
          "Fast Oct Tree Explosion", "Fast Oct Tree Cubeworld", "Fast Oct Tree Gold Cube",
          "Fast Oct Tree Apartment Building", "Fast Oct Tree Sinnoh", "One Sphere", "Fast Oct Tree Volcano"]
T_RTOL = 1e-5      # north star: within 1e-5 relative hit t and normals
N_ATOL = 1e-5


TIE_RTOL = 1e-9   # two different objects hit at the same t to this relative precision = degenerate tie


def check_ids(obj_gpu, t_gpu, obj_ref, t_ref):
    """Object ids must be bit-exact, except for degenerate ties: coplanar surfaces of two objects
    (Cornell smoke: the rotated box's bottom face lies in the floor plane) are hit at the same t,
    and the reference's own winner is then decided by the rounding noise of its arithmetic."""
    mism = obj_gpu != obj_ref
    both = (obj_gpu >= 0) & (obj_ref >= 0)
    tie = mism & both & (np.abs(t_gpu - t_ref) <= TIE_RTOL * np.abs(t_ref))
    hard = mism & ~tie
    assert hard.sum() == 0, f"{int(hard.sum())} of {len(obj_ref)} primary-hit ids differ (ties excluded: {int(tie.sum())})"
    assert tie.sum() <= 0.005 * len(obj_ref), f"{int(tie.sum())} ties"
    return ~mism


def gpu_render(rt, w, h, spp, first=0, x0=0, strip_w=None, accum=None):
    img = ParallelImage(accum) if accum is not None else ParallelImage.new_black(w, h)
    part = img.split(1)[0]
    part.offset_x, part.width = x0, (w - x0 if strip_w is None else strip_w)
    st = rt.trace_part(part, spp, first_sample=first)
    return img.buffer, st


@pytest.fixture(scope="module")
def worlds():
    cache = {}

    def get(name, seed=3):
        if (name, seed) not in cache:
            info = scene_by_name(name)
            rt = RayTracer(info, device=0, seed=seed)
            cache[(name, seed)] = (info, rt, oracle_ffi.OracleWorld(info))
        return cache[(name, seed)]

    return get


def test_device_is_visible():
    from ray_tracing_amd import _ffi
    assert _ffi.load().mrt_device_count() >= 1


@pytest.mark.parametrize("name", SCENES)
def test_bvh_bit_exact(worlds, name):
    info, rt, ow = worlds(name)
    a, b = rt.world.bvh(), ow.bvh()
    assert len(a) == len(b) and all((a[k] == b[k]).all() for k in a.dtype.names)


@pytest.mark.parametrize("name", SCENES)
def test_camera_rays(worlds, name):
    info, rt, ow = worlds(name)
    W, H = 97, 61
    ys, xs = np.mgrid[0:H, 0:W]
    px = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
    for sample in (0, 17):
        a = rt.world.camera_rays(W, H, px, 9, sample)
        b = ow.camera_rays(W, H, px, 9, sample)
        scale = np.abs(b[:, 3:6]).max()
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, scale)


@pytest.mark.parametrize("name", SCENES)
def test_tier_a_golden_rays(worlds, name):
    """Committed fixed rays -> committed checker answers."""
    info, rt, ow = worlds(name)
    g = np.load(os.path.join(GOLD, golden_key(name) + ".npz"))
    obj, t, normal, pos, ff = rt.world.trace_rays(g["rays"], seed=11)
    same = check_ids(obj, t, g["obj"], g["t"])
    assert (obj[:2048] == g["obj"][:2048]).all()     # camera (primary) rays: no ties, bit-exact
    hit = (g["obj"] >= 0) & same
    np.testing.assert_allclose(t[hit], g["t"][hit], rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(normal[hit], g["normal"][hit], rtol=0, atol=N_ATOL)
    np.testing.assert_allclose(pos[hit], g["pos"][hit], rtol=T_RTOL, atol=1e-6)
    assert (ff[hit] == g["ff"][hit]).all()


@pytest.mark.parametrize("name", SCENES)
def test_tier_a_many_rays_vs_live_checker(worlds, name):
    """>= 1e5 fixed rays per scene: camera rays plus random interior rays."""
    info, rt, ow = worlds(name)
    rays = probe_rays(ow, info, n_camera=60000, n_random=60000, seed=21)
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=5)
    oo, to, no, po, fo = ow.trace_rays(rays, seed=5)
    same = check_ids(og, tg, oo, to)
    assert (og[:60000] == oo[:60000]).all()          # camera (primary) rays: no ties, bit-exact
    hit = (oo >= 0) & same
    assert hit.sum() > 1000
    np.testing.assert_allclose(tg[hit], to[hit], rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(ng[hit], no[hit], rtol=0, atol=N_ATOL)
    assert (fg[hit] == fo[hit]).all()


def rectangle_border_rays(info, per_rect: int, seed: int) -> np.ndarray:
    """Rays from random interior origins aimed exactly at points on the borders (edges, corners, and points one
    ulp inside / outside them) of every untransformed rectangle of the scene, and axis-parallel rays (zero
    direction components) through them: where a rectangle's shape test and its box test are decided by rounding."""
    from ray_tracing_amd.scene import XYRect, XZRect, YZRect
    from tests.helpers import scene_bounds
    rng = np.random.default_rng(seed)
    c, r = scene_bounds(info)
    rays = []
    for ob in info.objects:
        sh = ob.shape
        if not isinstance(sh, (XYRect, XZRect, YZRect)) or not np.array_equal(ob.transform.world_transform, np.eye(4)):
            continue
        if isinstance(sh, XYRect):
            lo, hi, axes, flat = (sh.x0, sh.y0), (sh.x1, sh.y1), (0, 1), 2
        elif isinstance(sh, XZRect):
            lo, hi, axes, flat = (sh.x0, sh.z0), (sh.x1, sh.z1), (0, 2), 1
        else:
            lo, hi, axes, flat = (sh.y0, sh.z0), (sh.y1, sh.z1), (1, 2), 0
        for _ in range(per_rect):
            u = [lo[0] + rng.random() * (hi[0] - lo[0]), lo[1] + rng.random() * (hi[1] - lo[1])]
            side = int(rng.integers(0, 6))
            if side < 4:       # a point on one edge
                u[side // 2] = (lo, hi)[side % 2][side // 2]
            else:              # a corner
                u = [(lo, hi)[int(rng.integers(0, 2))][0], (lo, hi)[int(rng.integers(0, 2))][1]]
            nudge = int(rng.integers(0, 3))  # exactly on the border, or one ulp to either side
            if nudge:
                u[0] = np.nextafter(u[0], np.inf if nudge == 1 else -np.inf)
            target = np.zeros(3)
            target[axes[0]], target[axes[1]], target[flat] = u[0], u[1], sh.k
            origin = c + (rng.random(3) - 0.5) * r * 0.9
            if rng.random() < 0.25:  # an axis-parallel ray: two direction components are exactly zero
                origin = target.copy()
                origin[flat] = c[flat] + (rng.random() - 0.5) * r * 0.5
            d = target - origin
            if not np.any(d):
                continue
            rays.append(np.concatenate([origin, d * (0.5 + rng.random()), [0.0]]))
    # untransformed spheres: axis-parallel rays along the faces of the sphere's box (tangent to the sphere at the
    # point where it touches the face), exactly, one ulp inside and one ulp outside; and rays from random origins
    # through the six touching points
    from ray_tracing_amd.scene import Sphere
    for ob in info.objects:
        sh = ob.shape
        if not isinstance(sh, Sphere) or not np.array_equal(ob.transform.world_transform, np.eye(4)):
            continue
        ctr, rad = np.array(sh.origin, dtype=np.float64), abs(float(sh.radius))
        for _ in range(per_rect // 2):
            face_axis, run_axis = rng.permutation(3)[:2]
            touch = ctr.copy()
            touch[face_axis] += rad * (1.0 if rng.random() < 0.5 else -1.0)
            nudge = int(rng.integers(0, 3))
            if nudge:
                touch[face_axis] = np.nextafter(touch[face_axis], np.inf if nudge == 1 else -np.inf)
            if rng.random() < 0.6:
                origin = touch.copy()
                origin[run_axis] += (rng.random() - 0.5) * 6.0 * max(rad, 1.0)
                d = np.zeros(3)
                d[run_axis] = (touch[run_axis] - origin[run_axis]) or 1.0
            else:
                origin = c + (rng.random(3) - 0.5) * r * 0.9
                d = touch - origin
            rays.append(np.concatenate([origin, d * (0.5 + rng.random()), [0.0]]))
    return np.asarray(rays, dtype=np.float64).reshape(-1, 7)


@pytest.mark.parametrize("name", ["Cornell Smoke", "Easy Cornell Box", "Easy Scene", "Light Demonstration", "Two Sphere"])
def test_rays_at_rectangle_borders(worlds, name):
    """The wave walk takes an untransformed rectangle's or sphere's shape test for its box test (a rectangle's box
    is the rectangle, 0.002 thick, a sphere's touches it in six points); the per-lane walk tests every box. On borders, corners, one ulp beside them and along the axes
    the two walks -- same arithmetic otherwise -- must give the same answers (a ray lying IN a rectangle's plane
    meets it at t = 0 * inf = NaN, which passes every comparison of the shape test: there the box test decides and
    the wave walk still runs it). Against the checker the same rays
    agree except where a hit point computed with and without a fused multiply-add falls on different sides of a
    border (the documented contraction deviation): such rays hit the neighbour instead, and must stay a minority."""
    info, rt, ow = worlds(name)
    rays = rectangle_border_rays(info, per_rect=3000, seed=77)
    if len(rays) == 0:
        pytest.skip("no untransformed rectangle or sphere in this scene")
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=9)
    rt.world.set_option("traversal", 1)  # per-lane tree walk
    try:
        ol, tl, nl, pl, fl = rt.world.trace_rays(rays, seed=9)
    finally:
        rt.world.set_option("traversal", 0)
    # the walks may only differ where two rectangles meet: both hit at the same t (to rounding), and whether the
    # later one's box lets it replace the earlier one is decided by the last bit
    walk_mism = og != ol
    walk_tie = walk_mism & (og >= 0) & (ol >= 0) & (np.abs(tg - tl) <= TIE_RTOL * np.abs(tl))
    assert (walk_mism & ~walk_tie).sum() == 0, f"{int((walk_mism & ~walk_tie).sum())} of {len(ol)} ids differ between the walks beyond ties"
    assert walk_tie.sum() <= 0.005 * len(ol)
    same_walk = (ol >= 0) & ~walk_mism
    assert np.array_equal(tg[same_walk], tl[same_walk], equal_nan=True)  # (a ray lying in a rectangle's plane "hits" it at t = 0 * inf)
    oo, to, no, po, fo = ow.trace_rays(rays, seed=9)
    mism = og != oo
    # Measured (tests/border_rates.py -> profiles/r03_border_rates.json): of these rays, every one aimed exactly at a
    # border, the device and the checker disagree on 20.2 % (Cornell Smoke), 23.6 % (Easy Cornell Box), 9.6 % (Easy
    # Scene), 9.9 % (Light Demonstration), 9.5 % (Two Sphere); half of the Cornell disagreements are two surfaces at
    # the same t, a third a hit against a miss. The bound is the measured rate and a quarter. (With the rectangle's
    # t as the correctly rounded quotient and unfused in-plane coordinates the Cornell rate halves, to 10 %, for
    # 6.8 % of the headline kernel's speed -- measured, not kept.)
    bound = {"Cornell Smoke": 0.26, "Easy Cornell Box": 0.30, "Easy Scene": 0.125, "Light Demonstration": 0.125, "Two Sphere": 0.12}[name]
    print(f"{name}: border rays {len(rays)}, id mismatch against the checker {mism.mean():.4f} (bound {bound}), walks differ {int(walk_mism.sum())}")
    assert mism.mean() <= bound
    # (a medium's t hangs on the number of medium visits before it -- the draws are numbered by visit -- so one
    # border decision the other way moves it: compared for solid objects only)
    from ray_tracing_amd.scene import ConstantMedium
    solid = np.array([o >= 0 and not isinstance(info.objects[o].shape, ConstantMedium) for o in oo])
    hit = solid & ~mism
    assert hit.sum() > 0.4 * len(rays)
    np.testing.assert_allclose(tg[hit], to[hit], rtol=T_RTOL, atol=0)
    np.testing.assert_allclose(ng[hit], no[hit], rtol=0, atol=N_ATOL)


@pytest.mark.parametrize("name", ["Cornell Smoke", "Easy Scene", "Easy Cornell Box", "Fast Oct Tree Volume Ice", "Two Sphere"])
def test_wave_walk_equals_per_lane_walk(worlds, name):
    """The wave-synchronous walk has its own leaf code for media (boundary worked out once, second evaluation of a
    twice-evaluated medium from the same boundary hits) and rectangles; the per-lane walk runs the generic leaf test
    with its two-round loops. Same arithmetic, same draw numbering: ids and t must be equal bit for bit."""
    info, rt, ow = worlds(name)
    rays = probe_rays(ow, info, n_camera=30000, n_random=90000, seed=33)
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=13)
    rt.world.set_option("traversal", 1)
    try:
        ol, tl, nl, pl, fl = rt.world.trace_rays(rays, seed=13)
    finally:
        rt.world.set_option("traversal", 0)
    assert (og == ol).all(), f"{int((og != ol).sum())} ids differ"
    hit = ol >= 0
    assert hit.sum() > 10000
    assert np.array_equal(tg[hit], tl[hit])
    assert np.array_equal(ng[hit], nl[hit])


@pytest.mark.parametrize("name", SCENES)
def test_tier_p_golden_image(worlds, name):
    info, rt, ow = worlds(name, seed=3)
    g = np.load(os.path.join(GOLD, golden_key(name) + ".npz"))
    rt.world.set_option("trace_zero_throughput", 1)
    img, st = gpu_render(rt, 32, 32, 8)
    rt.world.set_option("trace_zero_throughput", 0)
    ref = g["image_32x32_spp8_seed3"]
    off = np.abs(img - ref).max(axis=2) > 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=2))
    assert off.sum() <= 3, f"{int(off.sum())} of 1024 pixels differ"
    assert abs(st.rays - int(g["rays_traced"])) <= 50 and abs(st.light_queries - int(g["light_queries"])) <= 50


@pytest.mark.parametrize("name", SCENES)
def test_tier_p_image_vs_live_checker(worlds, name):
    """Same seed, same sample indices: per-pixel agreement except rare discrete flips."""
    info, rt, ow = worlds(name, seed=3)
    W, H, spp = 160, 90, 16
    rt.world.set_option("trace_zero_throughput", 1)
    img, st = gpu_render(rt, W, H, spp, first=5)
    rt.world.set_option("trace_zero_throughput", 0)
    ref, rst = ow.render(W, H, spp, seed=3, first_sample=5, threads=16)
    tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=2))
    off = np.abs(img - ref).max(axis=2) > tol
    assert off.mean() < 2e-3, f"{int(off.sum())} of {W * H} pixels differ"
    assert abs(st.rays - rst.rays) <= max(8, rst.rays // 2000)   # a flipped zero-throughput walk is up to 49 rays
    assert st.paths == rst.paths == W * H * spp
    m_g, m_o = img.mean(axis=(0, 1), dtype=np.float64), ref.mean(axis=(0, 1), dtype=np.float64)
    np.testing.assert_allclose(m_g, m_o, rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("name", ["Cornell Smoke", "Random Scene", "Light Demonstration", "Voxel City"])
def test_tier_b_statistical_rmse(worlds, name):
    """RMSE protocol on linear pre-bloom radiance against an INDEPENDENT CPU estimator (mt19937_64)."""
    info, rt, ow = worlds(name, seed=3)
    # the headline scene at the protocol's size (480x270 at 1024 spp; two checker renders of about 21 s on 16 threads);
    # the other three at a size that keeps the suite short -- their protocol-size figures are on record in
    # profiles/r03_rmse.json (tests/rmse_record.py)
    W, H, spp = (480, 270, 1024) if name == "Cornell Smoke" else (96, 54, 256)
    img, _ = gpu_render(rt, W, H, spp)
    cpu_a, _ = ow.render(W, H, spp, seed=100, threads=16, rng_mode=1)
    cpu_b, _ = ow.render(W, H, spp, seed=200, threads=16, rng_mode=1)
    g, a, b = img / spp, cpu_a / spp, cpu_b / spp
    # display values clamp(sqrt(x)) (rgb_color.rs:75-80) damp fireflies
    disp = lambda x: np.clip(np.sqrt(np.maximum(x, 0)), 0, 1)
    r_cc, r_gc = rmse(disp(a), disp(b)), 0.5 * (rmse(disp(g), disp(a)) + rmse(disp(g), disp(b)))
    assert r_gc <= 1.15 * r_cc, (r_gc, r_cc)
    if name == "Cornell Smoke":   # radiance is bounded by the emitter here (no fireflies): the linear values too
        lin_cc, lin_gc = rmse(a, b), 0.5 * (rmse(g, a) + rmse(g, b))
        assert lin_gc <= 1.15 * lin_cc, (lin_gc, lin_cc)
        print(f"Cornell smoke {W}x{H} @ {spp}: display R_cc {r_cc:.6f} R_gc {r_gc:.6f}, linear R_cc {lin_cc:.6f} R_gc {lin_gc:.6f}")
    mg, mc = disp(g).mean(axis=(0, 1)), 0.5 * (disp(a).mean(axis=(0, 1)) + disp(b).mean(axis=(0, 1)))
    noise = np.abs(disp(a).mean(axis=(0, 1)) - disp(b).mean(axis=(0, 1)))
    assert (np.abs(mg - mc) <= 0.005 * mc + 2.0 * noise + 1e-4).all(), (mg, mc, noise)


def test_cornell_rays_equal_checker_when_walking_zero_throughput_paths(worlds):
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    rt.world.set_option("trace_zero_throughput", 1)
    _, st1 = gpu_render(rt, 64, 64, 8)
    rt.world.set_option("trace_zero_throughput", 0)
    _, st0 = gpu_render(rt, 64, 64, 8)
    _, rst = ow.render(64, 64, 8, seed=3, threads=8)
    assert abs(st1.rays - rst.rays) <= 50
    assert st0.rays <= st1.rays      # the black smoke (albedo 0) ends paths early, image unchanged


def test_early_termination_does_not_change_the_image(worlds):
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    rt.world.set_option("trace_zero_throughput", 1)
    a, _ = gpu_render(rt, 64, 64, 8)
    rt.world.set_option("trace_zero_throughput", 0)
    b, _ = gpu_render(rt, 64, 64, 8)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("opts", [dict(pool_slots=256, segments_per_launch=1), dict(pool_slots=4096, segments_per_launch=7),
                                  dict(pool_slots=1 << 20, segments_per_launch=50), dict(compaction=0),
                                  dict(traversal=1), dict(traversal=2), dict(traversal=3)])
def test_scheduling_options_do_not_change_results(opts):
    """Pool size / segments per launch / compaction only change the order of work."""
    info = scenarios.cornell_smoke()
    base = RayTracer(info, device=0, seed=12)
    ref, rst = gpu_render(base, 80, 45, 6)
    rt = RayTracer(info, device=0, seed=12)
    for k, v in opts.items():
        rt.world.set_option(k, v)
    img, st = gpu_render(rt, 80, 45, 6)
    assert st.rays == rst.rays and st.light_queries == rst.light_queries
    np.testing.assert_allclose(img, ref, rtol=1e-5, atol=1e-6)   # f32 atomic adds: order-dependent last bits


@pytest.mark.parametrize("name", ["Voxel City", "Fast Oct Tree Sinnoh", "Fast Oct Tree Sphere"])
def test_suspended_octree_walks_do_not_change_results(name):
    """Lanes whose octree walk outlasts most of their wave keep it for the next turn of the kernel loop ("octree_suspend",
    on by default). A ray takes the same steps whatever turn they are taken in: same counters, same image as with whole
    walks -- also when launches are so short (2 or 3 turns) that walks are put down at a launch's end and start again."""
    info = scene_by_name(name)
    whole = RayTracer(info, device=0, seed=12)
    whole.world.set_option("octree_suspend", 0)
    ref, rst = gpu_render(whole, 160, 90, 6)
    for opts in (dict(), dict(segments_per_launch=2), dict(segments_per_launch=3, pool_slots=4096)):
        rt = RayTracer(info, device=0, seed=12)
        for k, v in opts.items():
            rt.world.set_option(k, v)
        img, st = gpu_render(rt, 160, 90, 6)
        assert st.rays == rst.rays and st.light_queries == rst.light_queries and st.paths == rst.paths
        np.testing.assert_allclose(img, ref, rtol=1e-5, atol=1e-6)   # f32 atomic adds: order-dependent last bits


def test_strips_and_sample_ranges_compose(worlds):
    """Ragged strips and split sample ranges reproduce the one-call render (trace_part semantics)."""
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    W, H, spp = 83, 37, 5
    whole, _ = gpu_render(rt, W, H, spp)
    acc = np.zeros((H, W, 3), dtype=np.float32)
    for x0, sw in [(0, 1), (1, 40), (41, 42)]:
        gpu_render(rt, W, H, 2, first=0, x0=x0, strip_w=sw, accum=acc)
        gpu_render(rt, W, H, 3, first=2, x0=x0, strip_w=sw, accum=acc)
    np.testing.assert_allclose(acc, whole, rtol=1e-5, atol=1e-6)
    # a strip render leaves the other columns untouched
    acc2 = np.full((H, W, 3), 0.25, dtype=np.float32)
    gpu_render(rt, W, H, 1, x0=10, strip_w=5, accum=acc2)
    assert (acc2[:, :10] == 0.25).all() and (acc2[:, 15:] == 0.25).all()


def test_zero_spp_and_bad_arguments(worlds):
    from ray_tracing_amd import _ffi as F
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    img, st = gpu_render(rt, 8, 8, 0)
    assert (img == 0).all() and st.paths == 0
    lib = F.load()
    buf = np.zeros((8, 8, 3), dtype=np.float32)
    assert lib.mrt_render_accumulate(rt.world.handle, buf.ctypes.data, 8, 8, 4, 5, 1, 0, 0, None) == F.MRT_ERR_INVALID_ARGUMENT
    assert lib.mrt_render_accumulate(rt.world.handle, buf.ctypes.data, 1, 8, 0, 1, 1, 0, 0, None) == F.MRT_ERR_INVALID_ARGUMENT
    assert lib.mrt_render_accumulate(rt.world.handle, None, 8, 8, 0, 8, 1, 0, 0, None) == F.MRT_ERR_INVALID_ARGUMENT


def test_empty_world_renders_background():
    from ray_tracing_amd.scene import CameraInfo, Sky, WorldInfo
    info = WorldInfo([], [], Sky(1.0), CameraInfo(1.0, 40.0, (0, 0, 1), (0, 0, 0), (0, 1, 0), 0.0, 1.0, 0.0, 0.0))
    rt = RayTracer(info, device=0, seed=0)
    img, st = gpu_render(rt, 16, 16, 3)
    ref, rst = oracle_ffi.OracleWorld(info).render(16, 16, 3, seed=0, threads=1)
    assert st.rays == rst.rays == 16 * 16 * 3
    np.testing.assert_allclose(img, ref, rtol=1e-6, atol=1e-6)


def test_full_size_properties_1080p(worlds):
    """The benchmark's image size (1920x1080): properties that need no full-size CPU render."""
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    W, H = 1920, 1080
    a, sa = gpu_render(rt, W, H, 4, first=0)
    # (1) union of sample ranges == one call
    b, _ = gpu_render(rt, W, H, 1, first=0)
    gpu_render(rt, W, H, 3, first=1, accum=b)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    # (2) every path is accounted for, segments per path is within the depth limit
    assert sa.paths == W * H * 4 and sa.paths <= sa.rays <= 50 * sa.paths
    # (3) radiance is bounded by the emitter (7,7,7) per sample and never negative / NaN
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 7.0 * 4 * 1.0001
    # (4) image statistics agree with a low-resolution checker render of the same scene
    ref, _ = ow.render(240, 135, 32, seed=77, threads=16)
    np.testing.assert_allclose(a.mean(axis=(0, 1), dtype=np.float64) / 4, ref.mean(axis=(0, 1), dtype=np.float64) / 32,
                               rtol=0.03)
    # (5) the quirk columns: green wall / back wall black at full resolution
    assert a[400:700, 5:60].max() == 0.0 and a[500:560, 930:990].max() == 0.0


def test_to_image_matches_checker(worlds, oracle):
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    img, _ = gpu_render(rt, 64, 48, 4)
    got = ParallelImage(img).to_image(4)
    exp = np.zeros((48, 64, 3), dtype=np.uint8)
    oracle.oracle_to_image(img.ctypes.data, 64, 48, 4, exp.ctypes.data)
    assert (got == exp).all()


def test_full_size_properties_voxel_city_4k(worlds):
    """Voxel-city configuration at its stated size (3840x2160): size-independent properties."""
    info, rt, ow = worlds("Voxel City", seed=3)
    W, H = 3840, 2160
    a, sa = gpu_render(rt, W, H, 2, first=0)
    b, _ = gpu_render(rt, W, H, 1, first=0)
    gpu_render(rt, W, H, 1, first=1, accum=b)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)            # union of sample ranges
    assert sa.paths == W * H * 2 and sa.paths <= sa.rays <= 50 * sa.paths
    assert 0 < sa.light_queries <= sa.rays                            # LightPdf::value still queries the (empty) light list
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 22.0 * 2 * 1.0001   # sun_brightness bounds radiance
    # (means in f64: a float32 running sum over 8.3 M pixels loses several per cent)
    ref, _ = ow.render(240, 135, 32, seed=5, threads=16)
    np.testing.assert_allclose(a.mean(axis=(0, 1), dtype=np.float64) / 2, ref.mean(axis=(0, 1), dtype=np.float64) / 32,
                               rtol=0.02)
    # the sky above the skyline is the SunSky gradient: no geometry, exactly one ray per path there
    top = a[H - 40:, :400, :] / 2
    assert top.min() > 0.0 and top.max() < 0.06


def test_samples_split_over_scene_handles_in_one_process():
    """mrt_render_accumulate_multi: three handles (here on one GPU) render disjoint sample shares concurrently;
    the sum is the one-handle image up to the order of the f32 additions. 7 samples over 3 handles: 3 + 2 + 2."""
    info = scenarios.cornell_smoke()
    rts = [RayTracer(info, device=0, seed=21) for _ in range(3)]
    W, H, spp = 83, 47, 7
    ref, rst = gpu_render(rts[0], W, H, spp, first=5)
    img = ParallelImage(np.full((H, W, 3), 0.5, dtype=np.float32))
    part = img.split(1)[0]
    part.offset_x, part.width = 3, 70
    st = rts[0].trace_part_multi(rts[1:], part, spp, first_sample=5)
    assert st.paths == 70 * H * spp and st.launches >= 3
    want = np.full((H, W, 3), 0.5, dtype=np.float32)
    strip_ref, _ = gpu_render(rts[0], W, H, spp, first=5, x0=3, strip_w=70)
    want[:, 3:73] += strip_ref[:, 3:73]
    np.testing.assert_allclose(img.buffer, want, rtol=1e-5, atol=1e-6)
    assert (img.buffer[:, :3] == 0.5).all() and (img.buffer[:, 73:] == 0.5).all()
    # more handles than samples: the idle ones do nothing; the same handle twice is allowed (calls serialise)
    img2 = ParallelImage.new_black(W, H)
    st2 = rts[0].trace_part_multi([rts[0], rts[1], rts[2]], img2.split(1)[0], 2, first_sample=5)
    two, _ = gpu_render(rts[0], W, H, 2, first=5)
    assert st2.paths == W * H * 2
    np.testing.assert_allclose(img2.buffer, two, rtol=1e-5, atol=1e-6)
    from ray_tracing_amd import _ffi as F
    lib = F.load()
    assert lib.mrt_render_accumulate_multi(None, 0, img2.buffer.ctypes.data, W, H, 0, W, 1, 0, 0, None) == F.MRT_ERR_INVALID_ARGUMENT


def test_multi_handle_sum_through_rccl_world_of_one(monkeypatch):
    """The device-side join of mrt_render_accumulate_multi: with MRT_MULTI_FORCE_RCCL the per-device images go
    through ncclAllReduce even on one GPU (a communicator of one rank), and a failure of the collective is an
    error instead of a fall-back to the host sum."""
    monkeypatch.setenv("MRT_MULTI_FORCE_RCCL", "1")
    info = scenarios.cornell_smoke()
    rts = [RayTracer(info, device=0, seed=8) for _ in range(2)]
    W, H, spp = 64, 40, 5
    img = ParallelImage(np.full((H, W, 3), 0.25, dtype=np.float32))
    st = rts[0].trace_part_multi(rts[1:], img.split(1)[0], spp, first_sample=2)
    assert st.paths == W * H * spp
    monkeypatch.delenv("MRT_MULTI_FORCE_RCCL")
    ref, _ = gpu_render(rts[0], W, H, spp, first=2)
    np.testing.assert_allclose(img.buffer, ref + np.float32(0.25), rtol=1e-5, atol=1e-6)
    # twice in a row with another image size: the resident per-device image is re-made
    img2 = ParallelImage.new_black(50, 30)
    rts[0].trace_part_multi(rts[1:], img2.split(1)[0], 3, first_sample=0)
    ref2, _ = gpu_render(rts[0], 50, 30, 3, first=0)
    np.testing.assert_allclose(img2.buffer, ref2, rtol=1e-5, atol=1e-6)


def test_multi_handle_join_failure_modes(monkeypatch):
    """The two ways out of the device-side join. RCCL not there (nothing touched yet): the host sums the strips and
    the image is right. A failure after the collectives were enqueued: the images may be partly reduced, so the call
    is an error -- a host sum on top would count samples twice."""
    from ray_tracing_amd import _ffi as F
    info = scenarios.cornell_smoke()
    rts = [RayTracer(info, device=0, seed=8) for _ in range(2)]
    W, H, spp = 64, 40, 4
    ref, _ = gpu_render(rts[0], W, H, spp, first=1)
    monkeypatch.setenv("MRT_MULTI_FORCE_RCCL", "1")
    monkeypatch.setenv("MRT_MULTI_INJECT_FAILURE", "load")
    img = ParallelImage.new_black(W, H)
    st = rts[0].trace_part_multi(rts[1:], img.split(1)[0], spp, first_sample=1)
    assert st.paths == W * H * spp
    np.testing.assert_allclose(img.buffer, ref, rtol=1e-5, atol=1e-6)
    monkeypatch.setenv("MRT_MULTI_INJECT_FAILURE", "collective")
    lib = F.load()
    buf = np.zeros((H, W, 3), dtype=np.float32)
    handles = (C.c_void_p * 2)(rts[0].world.handle, rts[1].world.handle)
    rc = lib.mrt_render_accumulate_multi(handles, 2, buf.ctypes.data, W, H, 0, W, spp, 8, 1, None)
    assert rc == F.MRT_ERR_DEVICE and b"ncclAllReduce" in lib.mrt_last_error()
    monkeypatch.delenv("MRT_MULTI_INJECT_FAILURE")
    monkeypatch.delenv("MRT_MULTI_FORCE_RCCL")


def test_render_sharded_over_nccl_world_of_one(worlds):
    """The multi-process leg of the benchmark on the one GPU that is here: a process group over the "nccl" backend
    (RCCL) with one rank, this rank's share rendered into a device tensor, one all-reduce of that tensor."""
    import socket

    import torch
    import torch.distributed as dist

    from ray_tracing_amd.distributed import render_sharded
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    W, H, spp = 96, 54, 6
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        def render(first, count):
            rt.render_device(accum.data_ptr(), W, H, count, first_sample=first,
                             stream=torch.cuda.current_stream().cuda_stream)

        reduced = []

        def all_reduce():
            dist.all_reduce(accum)
            reduced.append(True)

        first, count = render_sharded(render, all_reduce, spp, 0, 1, first_sample=4)
        assert (first, count) == (4, spp) and not reduced      # one rank: nothing to join
        dist.all_reduce(accum)                                  # the collective itself, on the device buffer
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    ref, _ = gpu_render(rt, W, H, spp, first=4)
    np.testing.assert_allclose(accum.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)


def test_full_size_properties_cornell_4k(worlds):
    """One GPU's share of the eight-GPU configuration (Cornell smoke 3840x2160, samples split by index)."""
    from ray_tracing_amd.distributed import sample_range
    info, rt, ow = worlds("Cornell Smoke", seed=3)
    W, H = 3840, 2160
    a, sa = gpu_render(rt, W, H, 2, first=0)
    # two ranks' shares of a 2-sample render, accumulated into one buffer, are the one-call render
    b = np.zeros((H, W, 3), dtype=np.float32)
    for r in range(2):
        first, count = sample_range(2, r, 2)
        gpu_render(rt, W, H, count, first=first, accum=b)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    assert sa.paths == W * H * 2 and sa.paths <= sa.rays <= 50 * sa.paths
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 7.0 * 2 * 1.0001
    ref, _ = ow.render(240, 135, 32, seed=77, threads=16)
    np.testing.assert_allclose(a.mean(axis=(0, 1), dtype=np.float64) / 2, ref.mean(axis=(0, 1), dtype=np.float64) / 32,
                               rtol=0.03)
    assert a[800:1400, 10:120].max() == 0.0 and a[1000:1120, 1860:1980].max() == 0.0   # green wall / back wall black
