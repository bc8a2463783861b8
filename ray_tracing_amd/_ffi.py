"""ctypes binding of include/mrt.h (the C ABI of the HIP library).

Plumbing only: structure layouts, library loading and argument checking. The
library is required -- there is no CPU fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MRT_LIBRARY selects another build of the same ABI (e.g. the diagnostic libmrt_stamps.so)
LIB_PATH = os.environ.get("MRT_LIBRARY") or os.path.join(_HERE, "libmrt.so")

MRT_ABI_VERSION = 2

# status codes
MRT_OK = 0
MRT_ERR_INVALID_ARGUMENT = -1
MRT_ERR_UNSUPPORTED = -2
MRT_SHADER_RAYTRACING, MRT_SHADER_DIFFUSE, MRT_SHADER_LIGHT_MAP = 0, 1, 2
MRT_ERR_DEVICE = -3
MRT_ERR_NO_DEVICE = -4
MRT_ERR_OUT_OF_MEMORY = -5

# enums
TEX_SOLID, TEX_IMAGE, TEX_MULTIPLY, TEX_CHECKER, TEX_DEBUG_V, TEX_PERLIN = range(6)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT, MAT_ISOTROPIC = range(5)
(SHAPE_SPHERE, SHAPE_MOVING_SPHERE, SHAPE_XY_RECT, SHAPE_XZ_RECT, SHAPE_YZ_RECT, SHAPE_RENDER_BOX,
 SHAPE_CONSTANT_MEDIUM, SHAPE_OCTREE) = range(8)
OCT_EMPTY, OCT_LEAF, OCT_PARENT = range(3)
VOX_LAMBERTIAN, VOX_REFLECT, VOX_VOLUME = range(3)
VOX_EDGE_NONE, VOX_EDGE_LAMBERTIAN, VOX_EDGE_REFLECT = range(3)
BG_SKY, BG_SUN_SKY, BG_CONSTANT = range(3)


class Texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("image", C.c_int32),
                ("color", C.c_float * 3), ("reserved", C.c_int32)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_float))]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("fuzz", C.c_double),
                ("index_refraction", C.c_double), ("color", C.c_float * 3), ("reserved", C.c_int32)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("boundary", C.c_int32),
                ("octree", C.c_int32), ("p", C.c_double * 10)]


class Object(C.Structure):
    _fields_ = [("shape", C.c_int32), ("reserved", C.c_int32), ("world_transform", C.c_double * 16)]


class OctNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("size", C.c_uint32), ("first_child", C.c_int32), ("voxel", C.c_int32)]


class Voxel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", C.c_float * 3), ("fuzz", C.c_float), ("edge_kind", C.c_int32),
                ("edge_probability", C.c_float), ("edge_fuzz", C.c_float), ("edge_albedo", C.c_float * 3),
                ("reserved", C.c_int32), ("density", C.c_double)]


class Octree(C.Structure):
    _fields_ = [("root", C.c_uint32), ("n_nodes", C.c_uint32), ("nodes", C.POINTER(OctNode)),
                ("n_children", C.c_uint32), ("children", C.POINTER(C.c_uint32)),
                ("n_voxels", C.c_uint32), ("voxels", C.POINTER(Voxel))]


class CameraInfo(C.Structure):
    _fields_ = [("aspect_ratio", C.c_double), ("fov", C.c_double), ("origin", C.c_double * 3),
                ("look_at", C.c_double * 3), ("up_vector", C.c_double * 3), ("aperture", C.c_double),
                ("focus_distance", C.c_double), ("start_time", C.c_double), ("end_time", C.c_double)]


class Background(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("intensity", C.c_double),
                ("sun_radius", C.c_double), ("sun_theta", C.c_double), ("sun_phi", C.c_double),
                ("sun_brightness", C.c_double), ("color", C.c_float * 3), ("reserved2", C.c_int32)]


class Sun(C.Structure):
    _fields_ = [("has_sun", C.c_int32), ("reserved", C.c_int32), ("phi", C.c_double), ("theta", C.c_double),
                ("radius", C.c_double)]


class Perlin(C.Structure):
    _fields_ = [("ran_float", (C.c_double * 3) * 256), ("perm_x", C.c_uint32 * 256), ("perm_y", C.c_uint32 * 256),
                ("perm_z", C.c_uint32 * 256)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_textures", C.c_uint32), ("n_images", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_shapes", C.c_uint32), ("n_objects", C.c_uint32),
                ("n_lights", C.c_uint32), ("n_octrees", C.c_uint32),
                ("textures", C.POINTER(Texture)), ("images", C.POINTER(Image)),
                ("materials", C.POINTER(Material)), ("shapes", C.POINTER(Shape)),
                ("objects", C.POINTER(Object)), ("lights", C.POINTER(Object)),
                ("octrees", C.POINTER(Octree)),
                ("camera", CameraInfo), ("background", Background), ("sun", Sun),
                ("bvh_seed", C.c_uint64), ("n_perlins", C.c_uint32), ("reserved", C.c_uint32),
                ("perlins", C.POINTER(Perlin))]


class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_double * 3), ("bmax", C.c_double * 3), ("object", C.c_int32), ("skip", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("light_queries", C.c_uint64),
                ("launches", C.c_uint64), ("kernel_ms", C.c_double), ("total_ms", C.c_double),
                ("segment_kernel_avg_ms", C.c_double)]


# every symbol include/mrt.h declares
EXPORTS = [
    "mrt_abi_version", "mrt_device_count", "mrt_last_error", "mrt_scene_create", "mrt_scene_destroy",
    "mrt_scene_bvh", "mrt_camera_rays", "mrt_trace_rays", "mrt_render_accumulate",
    "mrt_render_accumulate_device", "mrt_render_accumulate_device_noalloc", "mrt_render_accumulate_multi", "mrt_to_image", "mrt_scene_set_option",
    "mrt_bloom_device", "mrt_bloom", "mrt_to_image_device",
    "mrt_voxels_create", "mrt_voxels_destroy", "mrt_voxels_set", "mrt_voxels_combine", "mrt_voxels_load_vox",
    "mrt_voxels_count", "mrt_voxels_linearise",
]

_lib = None


class MrtError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"mrt status {status}: {message}")
        self.status = status


def load() -> C.CDLL:
    """Loads libmrt.so (built by `make lib` / __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (hipcc --offload-arch=gfx950). "
                          "The HIP library is required; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_size_t
    lib.mrt_abi_version.restype = C.c_int
    lib.mrt_device_count.restype = C.c_int
    lib.mrt_last_error.restype = C.c_char_p
    lib.mrt_scene_create.restype = vp
    lib.mrt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_char_p, sz]
    lib.mrt_scene_destroy.restype = None
    lib.mrt_scene_destroy.argtypes = [vp]
    lib.mrt_scene_bvh.restype = C.c_int
    lib.mrt_scene_bvh.argtypes = [vp, C.POINTER(BvhNode), sz, C.POINTER(sz)]
    lib.mrt_camera_rays.restype = C.c_int
    lib.mrt_camera_rays.argtypes = [vp, u32, u32, vp, sz, u64, u64, vp]
    lib.mrt_trace_rays.restype = C.c_int
    lib.mrt_trace_rays.argtypes = [vp, vp, sz, u64, vp, vp, vp, vp, vp]
    lib.mrt_render_accumulate.restype = C.c_int
    lib.mrt_render_accumulate.argtypes = [vp, vp, u32, u32, u32, u32, u32, u64, u64, C.POINTER(Stats)]
    lib.mrt_render_accumulate_device.restype = C.c_int
    lib.mrt_render_accumulate_device.argtypes = [vp, vp, u32, u32, u32, u32, u32, u64, u64, vp, C.POINTER(Stats)]
    lib.mrt_render_accumulate_device_noalloc.restype = C.c_int
    lib.mrt_render_accumulate_device_noalloc.argtypes = [vp, vp, u32, u32, u32, u32, u32, u64, u64, vp, C.POINTER(Stats)]
    lib.mrt_render_accumulate_multi.restype = C.c_int
    lib.mrt_render_accumulate_multi.argtypes = [vp, u32, vp, u32, u32, u32, u32, u32, u64, u64, C.POINTER(Stats)]
    lib.mrt_to_image.restype = C.c_int
    lib.mrt_to_image.argtypes = [vp, u32, u32, u32, vp]
    lib.mrt_scene_set_option.restype = C.c_int
    lib.mrt_scene_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.mrt_bloom_device.restype = C.c_int
    lib.mrt_bloom_device.argtypes = [C.c_int, vp, u32, u32, u32, vp, vp]
    lib.mrt_bloom.restype = C.c_int
    lib.mrt_bloom.argtypes = [C.c_int, vp, u32, u32, u32, vp]
    lib.mrt_to_image_device.restype = C.c_int
    lib.mrt_to_image_device.argtypes = [C.c_int, vp, u32, u32, u32, vp, vp]
    lib.mrt_voxels_create.restype = vp
    lib.mrt_voxels_destroy.restype = None
    lib.mrt_voxels_destroy.argtypes = [vp]
    lib.mrt_voxels_set.restype = C.c_int
    lib.mrt_voxels_set.argtypes = [vp, vp, vp, sz, C.c_int]
    lib.mrt_voxels_combine.restype = C.c_int
    lib.mrt_voxels_combine.argtypes = [vp, vp, vp]
    lib.mrt_voxels_load_vox.restype = C.c_int
    lib.mrt_voxels_load_vox.argtypes = [vp, vp, sz]
    lib.mrt_voxels_count.restype = sz
    lib.mrt_voxels_count.argtypes = [vp]
    lib.mrt_voxels_linearise.restype = C.c_int
    lib.mrt_voxels_linearise.argtypes = [vp, C.POINTER(Octree)]
    if lib.mrt_abi_version() != MRT_ABI_VERSION:
        raise ImportError("libmrt.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != MRT_OK:
        raise MrtError(status, load().mrt_last_error().decode("utf-8", "replace"))
