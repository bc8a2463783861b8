"""The library's host-side octree builder (mrt_voxels_*, csrc/mrt_octree.cpp) against the numpy builder
(ray_tracing_amd/octree.py), which tests/test_octree.py ties to the literal port of the reference's
set / collapse code: same canonical tree for random voxel sets, overwrites, combine and .vox loading."""
import os

import numpy as np
import pytest

from ray_tracing_amd.octree import (NativeOctTree, OctTree, box_points, load_vox, parse_vox, solid_lambertian,
                                    solid_reflect, volume, write_vox)
from tests.test_octree import canonical

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def canon(vg):
    return canonical(vg.nodes, vg.children, vg.root, lambda v: tuple(np.float32(vg.records[v]).tolist()))


def both():
    return OctTree(), NativeOctTree()


def test_single_voxel_and_world_size():
    for pos in [(0, 0, 0), (5, 0, 0), (0, 63, 1), (64, 64, 64)]:
        a, b = both()
        for t in (a, b):
            t.set_rows(np.asarray([pos]), solid_lambertian((0.5, 0.25, 0.125)))
        ga, gb = a.linearise(), b.linearise()
        assert int(ga.nodes[ga.root]["size"]) == int(gb.nodes[gb.root]["size"])
        assert canon(ga) == canon(gb)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_sets_with_overwrites_and_mixed_voxel_kinds(seed):
    rng = np.random.default_rng(seed)
    rows = [solid_lambertian((0.1, 0.2, 0.3)), solid_lambertian((0.9, 0.9, 0.9)), solid_reflect((0.5, 0.5, 0.5), 0.25),
            volume(0.3, (0.5, 0.05, 0.5)), volume(0.3, (0.5, 0.05, 0.5), 0.6, solid_reflect((0.5, 0.05, 0.5), 0.3))]
    a, b = both()
    for _ in range(6):
        n = int(rng.integers(1, 400))
        coords = rng.integers(0, int(rng.choice([4, 16, 40])), size=(n, 3))
        pick = rng.integers(0, len(rows), size=n)
        r = np.stack([rows[i] for i in pick])
        for t in (a, b):
            t.set_rows(coords, r)
    # a solid block that collapses into large leaves, set with one shared record
    for t in (a, b):
        t.set_rows(box_points((8, 8, 8), (16, 16, 16)), rows[0])
    assert len(b) == len(a.voxel_rows()[0])
    assert canon(a.linearise()) == canon(b.linearise())


def test_overwritten_far_voxel_still_fixes_the_world_size():
    a, b = both()
    for t in (a, b):
        t.set_rows(np.asarray([(100, 3, 3)]), solid_lambertian((1.0, 0.0, 0.0)))
        t.set_rows(np.asarray([(1, 1, 1)]), solid_lambertian((0.0, 1.0, 0.0)))
    ga, gb = a.linearise(), b.linearise()
    assert int(gb.nodes[gb.root]["size"]) == 7 and canon(ga) == canon(gb)


def test_combine_with_offsets_and_dropped_negatives():
    a, b = both()
    oa, ob = both()
    pts = box_points((0, 0, 0), (5, 3, 4))
    for t in (oa, ob):
        t.set_rows(pts, solid_lambertian((0.2, 0.4, 0.6)))
        t.set_rows(np.asarray([(2, 2, 2)]), volume(0.5, (0.1, 0.1, 0.1)))
    for t in (a, b):
        t.set_rows(box_points((0, 0, 0), (3, 1, 3)), solid_lambertian((0.7, 0.7, 0.7)))
    a.combine(oa, (9, 1, 9))
    b.combine(ob, (9, 1, 9))
    a.combine(oa, (-2, 0, -1))      # part of it falls below zero and is dropped (combine.rs:11-15)
    b.combine(ob, (-2, 0, -1))
    assert canon(a.linearise()) == canon(b.linearise())


def test_load_vox_bytes_matches_python_loader():
    z = np.load(os.path.join(GOLD, "vox_2x2_vox.npz"))
    models, pos = [], 0
    for n in z["model_sizes"]:
        models.append(z["xyzi"][pos:pos + int(n)])
        pos += int(n)
    data = write_vox(models, z["palette"])
    vf = parse_vox(data)
    assert len(vf.models) == len(models) and all((a == b).all() for a, b in zip(vf.models, models))
    a = load_vox(vf)
    b = NativeOctTree().load_vox_bytes(data)
    assert canon(a.linearise()) == canon(b.linearise())


def test_city_tree_native_equals_numpy():
    """The 4K benchmark geometry (world/fast_oct_tree.rs:818-883) built by both builders."""
    from ray_tracing_amd import scenarios
    from ray_tracing_amd.octree import city_grid
    vf = scenarios.load_vox_fixture(os.path.join(GOLD, "skyscraper0_vox.npz"))
    a = city_grid(load_vox(vf))
    building = NativeOctTree().load_vox_bytes(write_vox(vf.models, vf.palette))
    c, rows = a.voxel_rows()            # same final voxel set through the native set path
    b = NativeOctTree()
    b.set_rows(c, rows)
    ga, gb = a.linearise(), b.linearise()
    assert canon(ga) == canon(gb)
    assert len(building) == len(load_vox(vf).voxel_rows()[0])


def test_errors():
    from ray_tracing_amd import _ffi as F
    b = NativeOctTree()
    with pytest.raises(F.MrtError):
        b.linearise()                   # empty tree has no bounding box
    with pytest.raises(F.MrtError):
        b.set_rows(np.asarray([(-1, 0, 0)]), solid_lambertian((1, 1, 1)))
    with pytest.raises(F.MrtError):
        b.load_vox_bytes(b"not a vox file at all....")


OCTREE_SCENARIOS = ["Fast Oct Tree Sphere", "Voxel City", "Fast Oct Tree 2 density", "Fast Oct Tree Volume Lambertian",
                    "Fast Oct Tree Volume Metal", "Fast Oct Tree Volume Ice", "Fast Oct Tree Explosion",
                    "Fast Oct Tree Cubeworld", "Fast Oct Tree Gold Cube", "Fast Oct Tree Apartment Building",
                    "Fast Oct Tree Sinnoh"]


@pytest.mark.parametrize("name", OCTREE_SCENARIOS)
def test_scenarios_build_the_same_tree_with_either_builder(name):
    """The scenarios build their octrees with the library's C++ builder (what the renders and goldens use); the
    numpy builder, which tests/test_octree.py ties to the literal port of the reference's set / collapse code,
    gives the same tree."""
    from ray_tracing_amd import scenarios
    from tests.helpers import scene_by_name
    assert scenarios._TREE_BUILDER == "native"
    native = scene_by_name(name).objects[0].shape
    scenarios.set_tree_builder("numpy")
    try:
        ref = scene_by_name(name).objects[0].shape
    finally:
        scenarios.set_tree_builder("native")
    assert canon(native) == canon(ref)


def test_volcano_tree_properties():
    """"Fast Oct Tree Volcano" (world/fast_oct_tree.rs:245-321), built by the C++ builder only (2.8e7 voxels): voxel
    count from the cone predicate by slices of constant y, world edge, and spot checks of the two materials."""
    from ray_tracing_amd import scenarios
    vg = scenarios.volcano().objects[0].shape
    assert int(vg.nodes[vg.root]["size"]) == 10                      # largest coordinate 799 -> 1024^3
    assert len(vg.records) == 2

    def lookup(x, y, z):
        i, size = vg.root, 10
        while vg.nodes[i]["kind"] == 2:
            size -= 1
            slot = (((x >> size) & 1) << 2) | (((y >> size) & 1) << 1) | ((z >> size) & 1)
            i = int(vg.children[int(vg.nodes[i]["first_child"]) + slot])
        return None if vg.nodes[i]["kind"] == 0 else vg.records[int(vg.nodes[i]["voxel"])][1:4]

    assert lookup(500, 299, 500) is None                              # inside the crater cone above y = 150
    rock, lava = lookup(300, 50, 500), lookup(500, 140, 500)
    np.testing.assert_allclose(rock, np.float32([0x5f, 0x15, 0x15]) / np.float32(255.0))
    np.testing.assert_allclose(lava, np.float32([255, 0xbb, 0]) / np.float32(255.0) * np.float32(80.0))
    assert lookup(500, 149, 500 + 70) is not None and lookup(0, 0, 0) is None and lookup(199, 0, 500) is None
    assert lookup(201, 0, 500) is not None and lookup(500, 0, 799) is not None and lookup(500, 0, 800) is None
