"""The C-ABI library loads, exports every symbol include/mrt.h declares, and rejects bad input
with error codes (no compute calls: these run without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ray_tracing_amd import _ffi as F
from ray_tracing_amd import scenarios
from ray_tracing_amd.scene import Object, Sphere, Lambertian, SolidColor, MovingSphere, WorldInfo, Sky, CameraInfo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = F.load()
    syms = declared_symbols()
    assert set(syms) == set(F.EXPORTS), (syms, F.EXPORTS)
    for s in syms:
        assert getattr(lib, s) is not None


def test_abi_version():
    assert F.load().mrt_abi_version() == F.MRT_ABI_VERSION


def test_struct_sizes_match_c_layout():
    # sizes implied by include/mrt.h on LP64
    assert C.sizeof(F.Texture) == 32
    assert C.sizeof(F.Material) == 40
    assert C.sizeof(F.Shape) == 96
    assert C.sizeof(F.Object) == 136
    assert C.sizeof(F.CameraInfo) == 15 * 8
    assert C.sizeof(F.BvhNode) == 56
    assert C.sizeof(F.Stats) == 56
    assert C.sizeof(F.OctNode) == 16
    assert C.sizeof(F.Voxel) == 56
    assert C.sizeof(F.Perlin) == 256 * 3 * 8 + 3 * 256 * 4


def _create(info, device=-1):
    lib = F.load()
    flat = info.flatten()
    err = C.create_string_buffer(256)
    h = lib.mrt_scene_create(C.byref(flat.desc), device, err, len(err))
    return lib, h, err.value.decode(), flat


def test_host_only_scene_create_and_destroy():
    lib, h, err, _ = _create(scenarios.cornell_smoke())
    assert h, err
    n = C.c_size_t(0)
    assert lib.mrt_scene_bvh(h, None, 0, C.byref(n)) == F.MRT_OK
    assert n.value == 19
    lib.mrt_scene_destroy(h)


def test_bad_abi_version_is_rejected():
    lib = F.load()
    flat = scenarios.one_sphere().flatten()
    flat.desc.abi_version = 99
    err = C.create_string_buffer(256)
    assert not lib.mrt_scene_create(C.byref(flat.desc), -1, err, len(err))
    assert b"ABI" in err.value


def test_null_descriptor_is_rejected():
    lib = F.load()
    err = C.create_string_buffer(256)
    assert not lib.mrt_scene_create(None, -1, err, len(err))
    assert err.value


def test_bad_indices_are_rejected():
    lib = F.load()
    flat = scenarios.one_sphere().flatten()
    flat.desc.shapes[0].material = 7
    err = C.create_string_buffer(256)
    assert not lib.mrt_scene_create(C.byref(flat.desc), -1, err, len(err))
    assert b"material" in err.value


def test_singular_transform_is_rejected():
    # hittable.rs:69-76: expect("transform is not invertible") -> error code, not an abort
    lib = F.load()
    flat = scenarios.one_sphere().flatten()
    for i in range(16):
        flat.desc.objects[0].world_transform[i] = 0.0
    flat.desc.objects[0].world_transform[15] = 1.0
    err = C.create_string_buffer(256)
    assert not lib.mrt_scene_create(C.byref(flat.desc), -1, err, len(err))
    assert b"invertible" in err.value


def test_moving_sphere_light_is_unsupported():
    # sphere.rs:187-192: prob()/generate_ray_in_area() are todo!() in the reference
    ms = MovingSphere((0, 0, 0), (0, 1, 0), 0.0, 1.0, 0.5, Lambertian(SolidColor((1, 1, 1))))
    info = WorldInfo([Object(ms)], [Object(ms)], Sky(1.0),
                     CameraInfo(1.0, 20.0, (3, 3, 2), (0, 0, -1), (0, 1, 0), 0.0, 5.0, 0.0, 0.0))
    lib, h, err, _ = _create(info)
    assert not h and "panic" in err


def test_render_without_device_image_fails_loudly():
    lib, h, err, _ = _create(scenarios.one_sphere())
    assert h
    buf = np.zeros((4, 4, 3), dtype=np.float32)
    rc = lib.mrt_render_accumulate(h, buf.ctypes.data, 4, 4, 0, 4, 1, 0, 0, None)
    assert rc == F.MRT_ERR_INVALID_ARGUMENT
    assert b"device" in lib.mrt_last_error()
    lib.mrt_scene_destroy(h)


def test_unknown_option_is_rejected():
    lib, h, err, _ = _create(scenarios.one_sphere())
    assert lib.mrt_scene_set_option(h, b"no_such_option", 1) == F.MRT_ERR_INVALID_ARGUMENT
    assert lib.mrt_scene_set_option(h, b"segments_per_launch", 0) == F.MRT_ERR_INVALID_ARGUMENT
    assert lib.mrt_scene_set_option(h, b"segments_per_launch", 4) == F.MRT_OK
    lib.mrt_scene_destroy(h)


def test_to_image_matches_reference_colour_mapping():
    # rgb_color.rs:75-80: clamp(sqrt(c)) * 255 rounded; parallel_image.rs:61: row y -> height-1-y
    lib = F.load()
    acc = np.zeros((2, 2, 3), dtype=np.float32)
    acc[0, 0] = [4.0, 1.0, 0.25]      # / 4 spp -> 1, .25, .0625 -> 255, 128 (127.5 rounds away), 64
    acc[1, 1] = [100.0, 0.0, 2.0]
    out = np.zeros((2, 2, 3), dtype=np.uint8)
    assert lib.mrt_to_image(acc.ctypes.data, 2, 2, 4, out.ctypes.data) == 0
    assert out[1, 0].tolist() == [255, 128, 64]
    assert out[0, 1].tolist() == [255, 0, 180]
