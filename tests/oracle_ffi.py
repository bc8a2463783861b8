"""ctypes binding of the CPU checker (oracle/libmrt_oracle.so). Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ray_tracing_amd import _ffi as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "libmrt_oracle.so")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_PATH):
        raise ImportError(f"{ORACLE_PATH} missing: run `make oracle`")
    lib = C.CDLL(ORACLE_PATH)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    lib.oracle_scene_create.restype = vp
    lib.oracle_scene_create.argtypes = [C.POINTER(F.SceneDesc), C.c_char_p, sz]
    lib.oracle_scene_destroy.argtypes = [vp]
    lib.oracle_scene_destroy.restype = None
    lib.oracle_scene_bvh.argtypes = [vp, C.POINTER(F.BvhNode), sz, C.POINTER(sz)]
    lib.oracle_camera_rays.argtypes = [vp, u32, u32, vp, sz, u64, u64, vp]
    lib.oracle_trace_rays.argtypes = [vp, vp, sz, u64, vp, vp, vp, vp, vp]
    lib.oracle_render_accumulate.argtypes = [vp, vp, u32, u32, u32, u32, u32, u64, u64, u32, C.c_int,
                                             C.POINTER(F.Stats)]
    lib.oracle_sample_radiance.argtypes = [vp, u32, u32, vp, sz, u64, u64, vp]
    lib.oracle_lambertian_generate.argtypes = [vp, vp, vp, vp, sz, u64, u32, vp, vp, vp, vp]
    lib.oracle_rng_u64.restype = u64
    lib.oracle_rng_u64.argtypes = [u64, u64, u64, u32, u32]
    lib.oracle_to_image.argtypes = [vp, u32, u32, u32, vp]
    lib.oracle_set_shader.argtypes = [vp, C.c_int]
    lib.oracle_bloom.argtypes = [vp, u32, u32, u32]
    lib.oracle_bloom.restype = C.c_int
    _lib = lib
    return lib


class OracleWorld:
    def __init__(self, info):
        self.lib = load()
        self.flat = info.flatten()
        err = C.create_string_buffer(512)
        self.handle = self.lib.oracle_scene_create(C.byref(self.flat.desc), err, len(err))
        if not self.handle:
            raise RuntimeError("oracle: " + err.value.decode())

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.oracle_scene_destroy(self.handle)
            self.handle = None

    def set_shader(self, shader: int) -> None:
        assert self.lib.oracle_set_shader(self.handle, shader) == 0

    def bvh(self):
        n = C.c_size_t(0)
        assert self.lib.oracle_scene_bvh(self.handle, None, 0, C.byref(n)) == 0
        nodes = (F.BvhNode * max(1, n.value))()
        assert self.lib.oracle_scene_bvh(self.handle, nodes, n.value, C.byref(n)) == 0
        return np.array([(list(x.bmin), list(x.bmax), x.object, x.skip) for x in nodes[:n.value]],
                        dtype=[("bmin", "f8", 3), ("bmax", "f8", 3), ("object", "i4"), ("skip", "i4")])

    def camera_rays(self, width, height, pixels, seed, sample_index):
        px = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((len(px), 7), dtype=np.float64)
        assert self.lib.oracle_camera_rays(self.handle, width, height, px.ctypes.data, len(px), seed, sample_index,
                                           out.ctypes.data) == 0
        return out

    def trace_rays(self, rays, seed=0):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        n = len(r)
        obj = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float64)
        normal = np.zeros((n, 3), dtype=np.float64)
        pos = np.zeros((n, 3), dtype=np.float64)
        ff = np.zeros(n, dtype=np.int32)
        assert self.lib.oracle_trace_rays(self.handle, r.ctypes.data, n, seed, obj.ctypes.data, t.ctypes.data,
                                          normal.ctypes.data, pos.ctypes.data, ff.ctypes.data) == 0
        return obj, t, normal, pos, ff

    def render(self, width, height, spp, seed=0, first_sample=0, x0=0, strip_w=None, threads=8, rng_mode=0,
               accum=None):
        if accum is None:
            accum = np.zeros((height, width, 3), dtype=np.float32)
        st = F.Stats()
        sw = width - x0 if strip_w is None else strip_w
        rc = self.lib.oracle_render_accumulate(self.handle, accum.ctypes.data, width, height, x0, sw, spp, seed,
                                               first_sample, threads, rng_mode, C.byref(st))
        assert rc == 0
        return accum, st

    def lambertian_generate(self, points, normals, in_times, seed, seg=0):
        """LambertianPDF::generate at given surface points: (direction n x 3, value n, branch n, some n)."""
        p = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        nr = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        tm = np.ascontiguousarray(in_times, dtype=np.float64).reshape(-1)
        n = len(p)
        d = np.zeros((n, 3), dtype=np.float64)
        val = np.zeros(n, dtype=np.float64)
        choice = np.zeros(n, dtype=np.int32)
        some = np.zeros(n, dtype=np.int32)
        assert self.lib.oracle_lambertian_generate(self.handle, p.ctypes.data, nr.ctypes.data, tm.ctypes.data, n, seed, seg,
                                                   d.ctypes.data, val.ctypes.data, choice.ctypes.data, some.ctypes.data) == 0
        return d, val, choice, some

    def sample_radiance(self, width, height, pixels, seed, sample_index):
        px = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((len(px), 3), dtype=np.float32)
        assert self.lib.oracle_sample_radiance(self.handle, width, height, px.ctypes.data, len(px), seed,
                                               sample_index, out.ctypes.data) == 0
        return out


def oracle_bloom(accum: np.ndarray, num_samples: int):
    """bloom(accum / num_samples) by the CPU restatement; None where the reference would panic."""
    img = np.ascontiguousarray(accum, dtype=np.float32).copy()
    h, w, _ = img.shape
    rc = load().oracle_bloom(img.ctypes.data, w, h, num_samples)
    return img if rc == 0 else None


def oracle_to_image(rgb: np.ndarray, num_samples: int) -> np.ndarray:
    img = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = img.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    assert load().oracle_to_image(img.ctypes.data, w, h, num_samples, out.ctypes.data) == 0
    return out
