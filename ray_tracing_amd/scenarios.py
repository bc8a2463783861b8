"""Scene constants of the reference's scenario constructors, as inputs.

Each function returns the `WorldInfo` that the reference's function of the same
name builds (cpu_rt/src/ray_tracer/world/*.rs); the reference registers them by
display name in `get_scenarios` (cpu_rt/src/ray_tracer/world.rs:153-269).
Where the reference draws scene parameters from its unseeded global RNG
(random_scene.rs:63-123), a seeded numpy generator is used instead.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np

from .scene import (CameraInfo, CheckerTexture, ConstantColor, ConstantMedium, DebugV, Dielectric, DiffuseLight,
                    ImageTexture, Isotropic, Lambertian, Metal, MovingSphere, MultiplyTexture, Object, Perlin, RenderBox,
                    Sky, SolidColor, Sphere, SunSky, Transform, WorldInfo, XYRect, XZRect, YZRect)


# Builder of the voxel octrees of the "Fast Oct Tree" scenarios: the library's C++ builder (`mrt_voxels_*`,
# what a compiled host calls) unless `set_tree_builder("numpy")` selects the numpy one (the cross-check).
_TREE_BUILDER = "native"


def set_tree_builder(kind: str) -> None:
    global _TREE_BUILDER
    if kind not in ("native", "numpy"):
        raise ValueError("tree builder must be 'native' or 'numpy'")
    _TREE_BUILDER = kind


def _tree_cls():
    from .octree import NativeOctTree, OctTree
    return NativeOctTree if _TREE_BUILDER == "native" else OctTree


def _dist(a, b) -> float:
    t = [a[i] - b[i] for i in range(3)]
    return math.sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2])


def cornell_smoke() -> WorldInfo:
    """world/cornell_smoke.rs:11-159. Note flip=False on the green wall, ceiling and back wall."""
    look_at, origin = (278.0, 278.0, 0.0), (278.0, 278.0, -800.0)
    green = Lambertian(SolidColor((0.12, 0.45, 0.15)))
    red = Lambertian(SolidColor((0.65, 0.05, 0.05)))
    light = DiffuseLight(SolidColor((7.0, 7.0, 7.0)))
    white = Lambertian(SolidColor((0.73, 0.73, 0.73)))
    top_light = Object(XZRect(113.0, 443.0, 127.0, 423.0, 554.0, light, True), Transform.identity())
    objects = [
        Object(YZRect(0.0, 555.0, 0.0, 555.0, 555.0, green, False)),
        Object(YZRect(0.0, 555.0, 0.0, 555.0, 0.0, red, False)),
        top_light,
        Object(XZRect(0.0, 555.0, 0.0, 555.0, 0.0, white, False)),
        Object(XZRect(0.0, 555.0, 0.0, 555.0, 555.0, white, False)),
        Object(XYRect(0.0, 555.0, 0.0, 555.0, 555.0, white, False)),
        Object(ConstantMedium(RenderBox((0.0, 0.0, 0.0), (165.0, 330.0, 165.0), white),
                              Isotropic(SolidColor((0.0, 0.0, 0.0))), 0.01),
               Transform.identity().translate((265.0, 0.0, 295.0))),
        Object(ConstantMedium(Sphere(100.0, (0.0, 0.0, 0.0), white), Isotropic(SolidColor((0.5, 0.0, 0.0))), 0.01),
               Transform.identity().translate((265.0, 500.0, 295.0))),
        Object(RenderBox((0.0, 0.0, 0.0), (165.0, 165.0, 165.0), white),
               Transform.identity().rotate_y(-18.0).translate((130.0, 0.0, 65.0))),
    ]
    return WorldInfo(
        objects=objects,
        lights=[Object(top_light.shape, Transform.identity())],
        background=ConstantColor((0.0, 0.0, 0.0)),
        camera=CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0),
        sun=None,
    )


def easy_cornell_box() -> WorldInfo:
    """world/easy_cornell_box.rs."""
    look_at, origin = (278.0, 278.0, 0.0), (278.0, 278.0, -800.0)
    green = Lambertian(SolidColor((0.12, 0.45, 0.15)))
    red = Lambertian(SolidColor((0.65, 0.05, 0.05)))
    light = DiffuseLight(SolidColor((15.0, 15.0, 15.0)))
    white = Lambertian(SolidColor((0.73, 0.73, 0.73)))
    top_light = Object(XZRect(213.0, 343.0, 227.0, 332.0, 554.0, light, True))
    objects = [
        Object(YZRect(0.0, 555.0, 0.0, 555.0, 555.0, green, True)),
        Object(YZRect(0.0, 555.0, 0.0, 555.0, 0.0, red, False)),
        Object(XZRect(0.0, 555.0, 0.0, 555.0, 0.0, white, False)),
        Object(XZRect(0.0, 555.0, 0.0, 555.0, 555.0, white, True)),
        Object(XYRect(0.0, 555.0, 0.0, 555.0, 555.0, white, True)),
        top_light,
    ]
    return WorldInfo(objects, [Object(top_light.shape)], ConstantColor((0.0, 0.0, 0.0)),
                     CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def one_sphere() -> WorldInfo:
    """world/one_sphere.rs (the reference's default scenario)."""
    look_at, origin = (0.0, 0.0, -1.0), (3.0, 3.0, 2.0)
    return WorldInfo(
        [Object(Sphere(0.5, (0.0, 0.0, -1.0), Lambertian(SolidColor((0.1, 0.2, 0.5)))))], [], Sky(1.0),
        CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def two_spheres() -> WorldInfo:
    """world/two_spheres.rs."""
    look_at, origin = (0.0, 0.0, -1.0), (3.0, 3.0, 2.0)
    return WorldInfo(
        [Object(Sphere(0.5, (0.0, 0.0, -1.0), Lambertian(SolidColor((0.1, 0.2, 0.5))))),
         Object(Sphere(0.5, (1.0, 0.0, -1.0), Metal(SolidColor((0.8, 0.6, 0.2)), 0.0)))], [], Sky(1.0),
        CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def random_scene(seed: int = 1) -> WorldInfo:
    """world/random_scene.rs:8-142: 22x22 grid of small spheres followed by the four big ones."""
    rng = np.random.default_rng(seed)
    r = lambda: float(rng.random())
    rf32 = lambda: float(np.float32(rng.random(dtype=np.float32)))
    objects = []
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = r()
            center = (a + 0.9 * r(), 0.2, b + 0.9 * r())
            if _dist(center, (4.0, 0.2, 0.0)) > 0.9:
                if choose_mat < 0.8:
                    c1 = (center[0], center[1] + r() * 0.5, center[2])
                    objects.append(Object(MovingSphere(center, c1, 0.0, 1.0, 0.2,
                                                       Lambertian(SolidColor((rf32(), rf32(), rf32()))))))
                elif choose_mat < 0.95:
                    albedo = SolidColor((rf32(), rf32(), rf32()))
                    objects.append(Object(Sphere(0.2, center, Metal(albedo, r() * 0.5 + 0.5))))
                else:
                    objects.append(Object(Sphere(0.2, center, Dielectric(1.5, (1.0, 1.0, 1.0)))))
    objects += [
        Object(Sphere(1000.0, (0.0, -1000.0, 1000.0), Lambertian(SolidColor((0.5, 0.5, 0.5))))),
        Object(Sphere(1.0, (0.0, 1.0, 0.0), Dielectric(1.5, (1.0, 1.0, 1.0)))),
        Object(Sphere(1.0, (-4.0, 1.0, 0.0), Lambertian(SolidColor((0.4, 0.2, 0.1))))),
        Object(Sphere(1.0, (4.0, 1.0, 0.0), Metal(SolidColor((0.4, 0.2, 0.1)), 0.0))),
    ]
    return WorldInfo(objects, [], Sky(1.0),
                     CameraInfo(1.0, 20.0, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.0005, 10.0, 0.0, 0.0))


def light_demo(earthmap: np.ndarray) -> WorldInfo:
    """world/light_demo.rs ("Light Demonstration"). `earthmap`: float32 (H, W, 3) decode of assets/earthmap.jpg."""
    look_at, origin = (0.0, 1.0, 0.0), (10.0, 10.0, 2.0)
    floor = Object(XZRect(-5.0, 5.0, -5.0, 5.0, 0.0, Lambertian(SolidColor((0.5, 0.5, 0.5))), False))
    l_sphere = Object(Sphere(1.0, (0.0, 1.0, 0.0), Metal(SolidColor((0.5, 0.1, 0.0)), 0.01)))
    light = Object(Sphere(0.2, (0.0, 3.0, 1.0),
                          DiffuseLight(MultiplyTexture(ImageTexture(earthmap), SolidColor((100.0, 100.0, 100.0))))))
    return WorldInfo([floor, light, l_sphere], [Object(light.shape)], Sky(0.3),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def easy_scene(earthmap: np.ndarray, perlin_seed: int = 0) -> WorldInfo:
    """world/easy_scene.rs:10-186 ("Easy Scene"): Checker(solid, Perlin) ground sphere, earthmap sphere, two
    Dielectric spheres (one with a negative radius), a DebugV metal sphere, and four lights of four shape
    kinds (XYRect, YZRect, RenderBox, Sphere) that are also in the object list. `Perlin::default()` draws
    its tables from the global RNG (texture.rs:162-180); here they come from `perlin_seed`."""
    look_at, origin = (0.0, 0.0, -1.0), (10.0, 3.0, 2.0)
    light = Object(XYRect(-0.5, 0.5, -0.5 + 1.0, 0.5 + 1.0, -2.3, DiffuseLight(SolidColor((0.0, 0.0, 0.5))), False))
    yz_light = Object(YZRect(-0.5, 0.5, -0.5, 0.5, -3.0, DiffuseLight(SolidColor((0.0, 0.5, 0.0))), False))
    box_light = Object(RenderBox((-0.2, -0.2 - 0.3, -0.2), (0.2, 0.2 - 0.3, 0.2),
                                 DiffuseLight(SolidColor((40000.0, 0.0, 0.0)))))
    sphere_light = Object(Sphere(0.5, (0.0, 1.5, -1.0), DiffuseLight(SolidColor((4.0, 4.0, 4.0)))))
    objects = [
        Object(Sphere(100.0, (0.0, -100.5, -1.0),
                      Lambertian(CheckerTexture(odd=Perlin.default(perlin_seed), even=SolidColor((0.5, 1.0, 0.0)))))),
        Object(Sphere(0.5, (0.0, 0.0, -1.0), Lambertian(ImageTexture(earthmap)))),
        Object(Sphere(0.5, (-1.0, 0.0, -1.0), Dielectric(1.5, (1.0, 0.8, 0.8)))),
        Object(Sphere(-0.45, (-1.0, 0.0, 1.0), Dielectric(1.5, (1.0, 1.0, 1.0)))),
        Object(Sphere(0.5, (1.0, 0.0, -1.0), Metal(DebugV(), 0.0))),
        light, yz_light, box_light, sphere_light,
    ]
    lights = [Object(light.shape), Object(yz_light.shape), Object(box_light.shape), Object(sphere_light.shape)]
    return WorldInfo(objects, lights, ConstantColor((0.0, 0.0, 0.0)),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def load_earthmap_fixture(path: str) -> np.ndarray:
    """The decoded assets/earthmap.jpg (tests/golden/earthmap_rgb8.npz, made by tests/golden/make_earthmap.py)
    as the reference stores it: float32 (H, W, 3), texel = u8 / 255, row 0 = top row (texture.rs:196-215)."""
    rgb8 = np.load(path)["rgb8"]
    return rgb8.astype(np.float32) / np.float32(255.0)


def synthetic_earthmap(width: int = 256, height: int = 128) -> np.ndarray:
    """A deterministic image with u8/255 texel values, for generated test scenes (not the reference's asset)."""
    y, x = np.mgrid[0:height, 0:width]
    r = ((x * 7 + y * 3) % 256).astype(np.float32) / np.float32(255.0)
    g = ((x * 2 + y * 11) % 256).astype(np.float32) / np.float32(255.0)
    b = ((x ^ y) % 256).astype(np.float32) / np.float32(255.0)
    return np.stack([r, g, b], axis=-1).astype(np.float32)


def fast_oct_tree_sphere() -> WorldInfo:
    """world/fast_oct_tree.rs:13-62 ("Fast Oct Tree Sphere"): VoxelGrid::sphere(10) lit by a sphere light."""
    from .octree import box_points
    OctTree = _tree_cls()
    radius = 10
    pts = box_points((0, 0, 0), (2 * radius + 2,) * 3)
    centre = np.array([radius, radius, radius], dtype=np.float64)
    dist = np.sqrt((((pts + 0.5) - centre) ** 2).sum(axis=1))      # shapes/mod.rs:4-27
    tree = OctTree()
    tree.set_many(pts[dist <= radius], np.float32([0.5, 0.5, 0.5]))
    origin, look_at = (-100.0, 10.0, 100.0), (10.0, 10.0, 10.0)
    light = Object(Sphere(10.0, (-320.0, 100.0, -100.0), DiffuseLight(SolidColor((400.0, 400.0, 400.0)))))
    return WorldInfo([Object(tree.linearise()), light], [Object(light.shape)], ConstantColor((0.1, 0.1, 0.1)),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def _volume_scene(tree, light_origin, light_power: float, background, origin=(-20.0, 5.0, -20.0)) -> WorldInfo:
    look_at = (5.0, 5.0, 5.0)
    light = Object(Sphere(1.0, light_origin, DiffuseLight(SolidColor((light_power,) * 3))))
    return WorldInfo([Object(tree.linearise()), light], [Object(light.shape)], background,
                     CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def _volume_block(tree, voxel_row) -> None:
    """The 10 x 8 x 10 block of volume voxels with a solid core that the volume scenarios share
    (world/fast_oct_tree.rs:346-388)."""
    from .octree import box_points, solid_lambertian
    red = solid_lambertian((0.65, 0.05, 0.05))
    pts = box_points((0, 1, 0), (10, 9, 10))
    tree.set_rows(pts, voxel_row(pts) if callable(voxel_row) else voxel_row)
    tree.set_rows(box_points((3, 3, 3), (6, 6, 6)), red)
    for p in [(0, 0, 0), (0, 1, 0), (5, 5, 5)]:
        tree.set_rows(np.asarray([p]), red)


def volume_two_density() -> WorldInfo:
    """world/fast_oct_tree.rs:322-407 ("Fast Oct Tree 2 density")."""
    from .octree import volume
    tree = _tree_cls()()

    def rows(pts):
        lo, hi = volume(0.3, (0.5, 0.05, 0.5)), volume(0.6, (0.5, 0.05, 0.5))
        return np.where((pts[:, 1] < 5)[:, None], lo[None, :], hi[None, :])

    _volume_block(tree, rows)
    return _volume_scene(tree, (50.0, 0.0, 20.0), 500.0, ConstantColor((0.1, 0.1, 0.1)))


def volume_lambertian() -> WorldInfo:
    """world/fast_oct_tree.rs:408-497 ("Fast Oct Tree Volume Lambertian")."""
    from .octree import solid_lambertian, volume
    tree = _tree_cls()()
    _volume_block(tree, volume(0.3, (0.5, 0.05, 0.5), 0.6, solid_lambertian((0.5, 0.05, 0.5))))
    return _volume_scene(tree, (50.0, 0.0, 20.0), 500.0, ConstantColor((0.1, 0.1, 0.1)))


def volume_metal() -> WorldInfo:
    """world/fast_oct_tree.rs:498-594 ("Fast Oct Tree Volume Metal")."""
    from .octree import box_points, solid_lambertian, solid_reflect, volume
    tree = _tree_cls()()
    _volume_block(tree, volume(0.3, (0.5, 0.05, 0.5), 0.6, solid_reflect((0.5, 0.05, 0.5), 0.3)))
    tree.set_rows(box_points((0, 0, 0), (40, 1, 40)), solid_lambertian((0.9, 0.9, 0.9)))
    return _volume_scene(tree, (0.0, 100.0, 200.0), 1000.0, ConstantColor((0.1, 0.1, 0.1)))


def volume_ice() -> WorldInfo:
    """world/fast_oct_tree.rs:595-690 ("Fast Oct Tree Volume Ice")."""
    from .octree import box_points, from_color_hex, solid_lambertian, solid_reflect, volume
    ice, snow = from_color_hex("#06068dff"), from_color_hex("#ffffffff")
    tree = _tree_cls()()
    pts = box_points((0, 1, 0), (10, 9, 10))
    snow_row = volume(0.001, snow, 1.0, solid_lambertian(snow))
    ice_row = volume(0.5, ice, 0.7, solid_reflect(ice, 0.1))
    tree.set_rows(pts + np.array([10, 0, 10]), np.where((pts[:, 1] >= 8)[:, None], snow_row[None, :], ice_row[None, :]))
    red = solid_lambertian((0.65, 0.05, 0.05))
    tree.set_rows(np.asarray([(0, 0, 0)]), red)
    tree.set_rows(np.asarray([(0, 1, 0)]), red)
    floor = box_points((0, 0, 0), (40, 0, 40))
    checker = ((floor[:, 0] + floor[:, 2]) & 1) == 0
    tree.set_rows(floor, np.where(checker[:, None], solid_lambertian(from_color_hex("#ffffffff"))[None, :],
                                  solid_lambertian(from_color_hex("#fc0202ff"))[None, :]))
    return _volume_scene(tree, (0.0, 100.0, 200.0), 500000.0, Sky(0.6), origin=(-20.0, 20.0, -20.0))


def load_vox_fixture(path: str):
    """Reads a decoded .vox fixture (tests/golden/*_vox.npz) back into a VoxFile."""
    from .octree import VoxFile
    z = np.load(path)
    models, pos = [], 0
    for n in z["model_sizes"]:
        models.append(z["xyzi"][pos:pos + int(n)])
        pos += int(n)
    return VoxFile(models, z["palette"])


def voxel_city(building_vox, scale_roads: bool = True) -> WorldInfo:
    """Voxel-city configuration: geometry of `fast_oct_tree::city()` (world/fast_oct_tree.rs:818-883, built from
    voxel_assets/city/skyscraper0.vox) with the camera and SunSky stored in examples/voxel_city_big.mscene
    (the .mscene itself holds no geometry; its loader sets sun = None and no lights, save_file.rs:108-131)."""
    from .octree import city_grid, load_vox
    grid = city_grid(load_vox(building_vox, _tree_cls()))
    bg = SunSky(intensity=0.05, sun_radius=0.08726646259971647, sun_theta=1.231, sun_phi=0.5235987755982988,
                sun_brightness=22.0)
    cam = CameraInfo(1.0, 40.0, (-150.0, 200.0, -150.0), (512.0, 10.0, 512.0), (0.0, 1.0, 0.0), 1e-05,
                     955.2947189218623, 0.0, 0.0)
    return WorldInfo([Object(grid.linearise())], [], bg, cam, sun=None)


def explosion() -> WorldInfo:
    """world/fast_oct_tree.rs:103-172 ("Fast Oct Tree Explosion"): a cosine height field of two colours under an
    orange sphere light. The heights are f32 arithmetic there; numpy's f32 cos may differ from Rust's in the last
    bit, which could move a column by one voxel where the height is within rounding of an integer."""
    xs, zs = np.meshgrid(np.arange(100), np.arange(100), indexing="ij")
    radius = np.sqrt((np.float32(50.0) - xs.astype(np.float32)) ** 2 + (np.float32(50.0) - zs.astype(np.float32)) ** 2,
                     dtype=np.float32)
    h = np.cos(radius / np.float32(10.0), dtype=np.float32) * np.float32(10.0) + np.float32(15.0)
    h = np.minimum(np.maximum(h, np.float32(0.0)), np.float32(19.0)).astype(np.int64)      # `as isize` truncates
    tree = _tree_cls()()
    cols = [(x, y, z) for x in range(100) for z in range(100) for y in range(int(h[x, z]) + 1)]
    pts = np.asarray(cols, dtype=np.int64)
    albedo = np.where((pts[:, 1] < 9)[:, None], np.float32([0.65, 0.8, 0.05])[None, :], np.float32([0.65, 0.05, 0.05])[None, :])
    tree.set_many(pts, albedo)
    origin, look_at = (-20.0, 50.0, -20.0), (50.0, 10.0, 50.0)
    e = [float(np.float32(20000.0) * (np.float32(c) / np.float32(255.0))) for c in (252.0, 79.0, 5.0)]
    light = Object(Sphere(3.0, (50.0, 28.0, 50.0), DiffuseLight(SolidColor(tuple(e)))))
    return WorldInfo([Object(tree.linearise()), light], [Object(light.shape)], Sky(0.1),
                     CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def cube_world() -> WorldInfo:
    """world/fast_oct_tree.rs:173-244 ("Fast Oct Tree Cubeworld"): a 3 x 3 x 3 block and three single voxels."""
    from .octree import box_points
    tree = _tree_cls()()
    red = np.float32([0.65, 0.05, 0.05])
    tree.set_many(box_points((3, 3, 3), (6, 6, 6)), red)
    tree.set_many(np.asarray([(0, 0, 0), (0, 1, 0), (5, 5, 5)]), red)
    origin, look_at = (-20.0, 5.0, -20.0), (0.0, 5.0, 5.0)
    light = Object(Sphere(1.0, (10.0, 10.0, -10.0), DiffuseLight(SolidColor((20000.0, 20000.0, 20000.0)))))
    return WorldInfo([Object(tree.linearise()), light], [Object(light.shape)], ConstantColor((0.1, 0.1, 0.1)),
                     CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def volcano() -> WorldInfo:
    """world/fast_oct_tree.rs:245-321 ("Fast Oct Tree Volcano"): every cell of a 1000^3 sweep inside a cone
    (apex (500, 300, 500), radius 300 - y) and not inside the crater cone (apex (500, 125, 500), radius
    (y - 125) / 0.3) unless below y = 150; cells inside both are lava (emissive-bright albedo #ffbb00 x 80), the
    rest rock #5f1515. About 2.8e7 voxels, world 1024^3, lit by a sphere light that is also the light list.
    The cone predicate is evaluated slice by slice in f64 exactly as the reference writes it."""
    from .octree import from_color_hex, solid_lambertian
    tree = _tree_cls()()
    lava = solid_lambertian(tuple(np.float32(c) * np.float32(80.0) for c in np.float32(from_color_hex("#ffbb00"))))
    rock = solid_lambertian(from_color_hex("#5f1515"))

    def cone(px, py, pz, center, slope):
        desired_radius = (py - center[1]) / slope
        r2 = np.where(desired_radius > 0.0, desired_radius * desired_radius, 0.0)
        return (px - center[0]) ** 2 + (pz - center[2]) ** 2 < r2

    # the mountain's radius never exceeds 300 and it ends below y = 300: only that part of the sweep can be set
    z0 = 200
    py, pz = np.meshgrid(np.arange(300, dtype=np.float64), np.arange(z0, 801, dtype=np.float64), indexing="ij")
    for x in range(200, 801):                  # IterBox order: x outermost (prelude.rs:57-77)
        px = float(x)
        mountain = cone(px, py, pz, (500.0, 300.0, 500.0), -1.0)
        crater = cone(px, py, pz, (500.0, 125.0, 500.0), 0.3)
        keep = mountain & (~crater | (py < 150.0))
        for mask, row in ((keep & crater, lava), (keep & ~crater, rock)):
            yy, zz = np.nonzero(mask)
            if len(yy):
                tree.set_rows(np.stack([np.full(len(yy), x), yy, zz + z0], axis=1), row)
    origin, look_at = (1000.0, 500.0, -300.0), (500.0, 200.0, 500.0)
    light = Object(Sphere(10.0, (-100.0, 50.0, -100.0), DiffuseLight(SolidColor((100.0, 100.0, 100.0)))))
    return WorldInfo([Object(tree.linearise()), light], [Object(light.shape)], ConstantColor((0.1, 0.1, 0.1)),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def gold_cube() -> WorldInfo:
    """world/fast_oct_tree.rs:692-786 ("Fast Oct Tree Gold Cube"): three reflecting 16^3 cubes of fuzz 0.8, 0.4 and
    0.1 on a 128 x 128 floor."""
    from .octree import box_points, solid_lambertian, solid_reflect
    tree = _tree_cls()()
    tree.set_rows(box_points((0, 0, 0), (128, 1, 128)), solid_lambertian((0.8, 0.8, 0.8)))
    gold = (1.0, float(np.float32(254.0) / np.float32(255.0)), 0.0)
    for x, fuzz in [(30, 0.8), (60, 0.4), (90, 0.1)]:
        tree.set_rows(box_points((x, 1, 64), (x + 16, 17, 80)), solid_reflect(gold, fuzz))
    origin, look_at = (64.0, 100.0, -300.0), (64.0, 0.0, 64.0)
    light = Object(Sphere(10.0, (-100.0, 50.0, -100.0), DiffuseLight(SolidColor((100.0, 100.0, 100.0)))))
    return WorldInfo([Object(tree.linearise()), light], [Object(light.shape)], ConstantColor((0.1, 0.1, 0.1)),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def apartment_building(building_vox) -> WorldInfo:
    """world/fast_oct_tree.rs:787-817 ("Fast Oct Tree Apartment Building"): voxel_assets/apartment_building.vox
    under Sky(0.4), no lights."""
    from .octree import load_vox
    origin, look_at = (-50.0, 100.0, -40.0), (30.0, 10.0, 30.0)
    return WorldInfo([Object(load_vox(building_vox, _tree_cls()).linearise())], [], Sky(0.4),
                     CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def load_map_fixture(path: str):
    """Reads a decoded tile map (tests/golden/sinnoh_map.npz: the .yml's tile table and the .vox models it names)
    back into (tiles, tile_size, {tile type: VoxFile})."""
    from .octree import VoxFile
    z = np.load(path)
    models = {}
    for k in z["model_types"]:
        sizes, xyzi, pos, ms = z[f"m{int(k)}_sizes"], z[f"m{int(k)}_xyzi"], 0, []
        for n in sizes:
            ms.append(xyzi[pos:pos + int(n)])
            pos += int(n)
        models[int(k)] = VoxFile(ms, z[f"m{int(k)}_palette"])
    return z["tiles"], int(z["tile_size"]), models


def sinnoh(tiles, tile_size: int, models) -> WorldInfo:
    """world/fast_oct_tree.rs:64-102 ("Fast Oct Tree Sinnoh"): the 32 x 32 tile map voxel_assets/sinnoh/twinleaf.yml
    under Sky(0.6), no lights."""
    from .octree import load_map
    origin, look_at = (-500.0, 300.0, 256.0), (256.0, 10.0, 256.0)
    return WorldInfo([Object(load_map(tiles, tile_size, models, _tree_cls()).linearise())], [], Sky(0.6),
                     CameraInfo(1.0, 40.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def new_demo(special_items) -> WorldInfo:
    """world/demo.rs:12-67: a grey floor, a small bright sphere light (also the light list), the items, Sky(1)."""
    look_at, origin = (0.0, 1.0, 0.0), (10.0, 10.0, 2.0)
    floor = Object(XZRect(-5.0, 5.0, -5.0, 5.0, 0.0, Lambertian(SolidColor((0.5, 0.5, 0.5))), False))
    light = Object(Sphere(0.2, (0.0, 3.0, 1.0), DiffuseLight(SolidColor((2000.0, 2000.0, 2000.0)))))
    return WorldInfo([floor, light] + list(special_items), [Object(light.shape)], Sky(1.0),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def lambertian_demo() -> WorldInfo:
    """world/demo/lambertian.rs ("Lambertian Demonstration")."""
    return new_demo([Object(Sphere(1.0, (0.0, 1.0, 0.0), Lambertian(SolidColor((0.5, 0.1, 0.0)))))])


def metallic_demo(fuzz: float) -> WorldInfo:
    """world/demo/metalic_demo.rs ("Metallic Demonstration Smooth": fuzz 0, "... Rough": fuzz 0.8)."""
    return new_demo([Object(Sphere(1.0, (0.0, 1.0, 0.0), Metal(SolidColor((0.5, 0.1, 0.0)), fuzz)))])


def cube_field() -> WorldInfo:
    """world/demo/cube_field.rs ("Cube Field"): two unit boxes, translated then rotated about y."""
    boxes = [Object(RenderBox((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), Lambertian(SolidColor((0.5, 0.1, 0.0)))),
                    Transform.identity().translate((x, 0.5, z)).rotate_y(deg))
             for x, z, deg in [(-0.5, -0.6, 20.0), (0.6, 0.6, -2.0)]]
    return new_demo(boxes)


def dielectric_demo(refraction: float) -> WorldInfo:
    """world/dielectric.rs ("Dielectric Demonstration, Low Refraction": 1.0, "... High Refraction": 1.5).
    The camera's field of view is 1 degree there."""
    look_at, origin = (0.0, 1.0, 0.0), (10.0, 10.0, 2.0)
    floor = Object(XZRect(-5.0, 5.0, -5.0, 5.0, 0.0, Lambertian(SolidColor((0.5, 0.5, 0.5))), False))
    glass = Object(Sphere(1.0, (0.0, 1.0, 0.0), Dielectric(refraction, (0.8, 0.8, 0.8))))
    behind = Object(Sphere(1.0, (-1.0, 1.0, 1.0), Lambertian(SolidColor((0.5, 0.1, 0.0)))))
    light = Object(Sphere(0.2, (0.0, 3.0, 1.0), DiffuseLight(SolidColor((2.0, 2.0, 2.0)))))
    return WorldInfo([floor, light, behind, glass], [Object(light.shape)], Sky(0.3),
                     CameraInfo(1.0, 1.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


def empty_scene() -> WorldInfo:
    """world/empty_scene.rs ("Empty Scene"): no objects, black background."""
    look_at, origin = (0.0, 0.0, -1.0), (10.0, 3.0, 2.0)
    return WorldInfo([], [], ConstantColor((0.0, 0.0, 0.0)),
                     CameraInfo(1.0, 20.0, origin, look_at, (0.0, 1.0, 0.0), 0.00001, _dist(look_at, origin), 0.0, 0.0))


SCENARIOS: Dict[str, Callable[[], WorldInfo]] = {
    "Lambertian Demonstration": lambertian_demo,
    "Metallic Demonstration Smooth": lambda: metallic_demo(0.0),
    "Metallic Demonstration Rough": lambda: metallic_demo(0.8),
    "Dielectric Demonstration, Low Refraction": lambda: dielectric_demo(1.0),
    "Dielectric Demonstration, High Refraction": lambda: dielectric_demo(1.5),
    "Cube Field": cube_field,
    "Empty Scene": empty_scene,
    "Fast Oct Tree Explosion": explosion,
    "Fast Oct Tree Cubeworld": cube_world,
    "Fast Oct Tree Gold Cube": gold_cube,
    "Fast Oct Tree Sphere": fast_oct_tree_sphere,
    "Fast Oct Tree Volcano": volcano,
    "Cornell Smoke": cornell_smoke,
    "Easy Cornell Box": easy_cornell_box,
    "One Sphere": one_sphere,
    "Two Sphere": two_spheres,
    "Random Scene": random_scene,
    "Fast Oct Tree 2 density": volume_two_density,
    "Fast Oct Tree Volume Lambertian": volume_lambertian,
    "Fast Oct Tree Volume Metal": volume_metal,
    "Fast Oct Tree Volume Ice": volume_ice,
}


def build(name: str, earthmap: Optional[np.ndarray] = None) -> WorldInfo:
    """Scenario by its display name. The two scenarios with an image texture need the decoded
    assets/earthmap.jpg (`load_earthmap_fixture`)."""
    if name in ("Light Demonstration", "Easy Scene"):
        if earthmap is None:
            raise ValueError(f"{name!r} needs the decoded earthmap image (scenarios.load_earthmap_fixture)")
        return light_demo(earthmap) if name == "Light Demonstration" else easy_scene(earthmap)
    return SCENARIOS[name]()


ASSET_SCENARIOS = ("Light Demonstration", "Easy Scene", "Voxel City", "Fast Oct Tree Apartment Building",
                   "Fast Oct Tree Sinnoh")


def build_with_assets(name: str, assets_dir: str) -> WorldInfo:
    """Scenario by its display name, for a host that holds the reference's asset files as decoded data:
    `assets_dir` contains `earthmap_rgb8.npz` (assets/earthmap.jpg), `skyscraper0_vox.npz`,
    `apartment_building_vox.npz` (voxel_assets/**.vox) and `sinnoh_map.npz` (voxel_assets/sinnoh/twinleaf.yml
    with its models). Scenarios without assets ignore the directory."""
    import os
    if name == "Voxel City":
        return voxel_city(load_vox_fixture(os.path.join(assets_dir, "skyscraper0_vox.npz")))
    if name == "Fast Oct Tree Apartment Building":
        return apartment_building(load_vox_fixture(os.path.join(assets_dir, "apartment_building_vox.npz")))
    if name == "Fast Oct Tree Sinnoh":
        return sinnoh(*load_map_fixture(os.path.join(assets_dir, "sinnoh_map.npz")))
    if name in ("Light Demonstration", "Easy Scene"):
        return build(name, load_earthmap_fixture(os.path.join(assets_dir, "earthmap_rgb8.npz")))
    return build(name)
