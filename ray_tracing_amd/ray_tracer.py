"""Host mirror of `RayTracer` / `ParallelImage` over the C ABI.

Names follow the reference (cpu_rt/src/ray_tracer.rs:330-580,
cpu_rt/src/parallel_image.rs:19-126,241-332): `trace_image` adds one sample per
pixel, `tracing_loop` adds `num_samples`, `trace_part` works on one vertical
strip. All pixel work happens in libmrt.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import _ffi as F
from .scene import WorldInfo


class ParallelImage:
    """f32 RGB accumulation image, index x + y*width, row y = 0 at the bottom (parallel_image.rs:19-24)."""

    def __init__(self, buffer: np.ndarray):
        assert buffer.dtype == np.float32 and buffer.ndim == 3 and buffer.shape[2] == 3
        self.buffer = np.ascontiguousarray(buffer)

    @staticmethod
    def new_black(width: int, height: int) -> "ParallelImage":
        return ParallelImage(np.zeros((height, width, 3), dtype=np.float32))

    @property
    def width(self) -> int:
        return self.buffer.shape[1]

    @property
    def height(self) -> int:
        return self.buffer.shape[0]

    def get_xy(self, x: int, y: int) -> np.ndarray:
        return self.buffer[y, x]

    def __truediv__(self, rhs: float) -> "ParallelImage":  # parallel_image.rs:268-278
        return ParallelImage(self.buffer / np.float32(rhs))

    def split(self, num_parts: int) -> List["ParallelImagePart"]:
        """Vertical strips of width // num_parts columns; the remainder columns are dropped
        exactly as the reference does (parallel_image.rs:103-126)."""
        slice_width = self.width // num_parts
        parts = []
        for s in range(num_parts):
            start = slice_width * s
            end = min(start + slice_width, self.width)
            parts.append(ParallelImagePart(self, start, end - start))
        return parts

    def to_image(self, num_samples: int) -> np.ndarray:
        """uint8 (height, width, 3), top row first (parallel_image.rs:35-65)."""
        lib = F.load()
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        F.check(lib.mrt_to_image(self.buffer.ctypes.data, self.width, self.height, num_samples, out.ctypes.data))
        return out


@dataclass
class ParallelImagePart:
    """A vertical strip view: columns [offset_x, offset_x + width) of `image` (parallel_image.rs:279-332)."""
    image: ParallelImage
    offset_x: int
    width: int

    @property
    def height(self) -> int:
        return self.image.height

    @property
    def total_width(self) -> int:
        return self.image.width


@dataclass
class RenderStats:
    paths: int
    rays: int
    light_queries: int
    launches: int
    kernel_ms: float
    total_ms: float
    segment_kernel_avg_ms: float


class World:
    """Device-resident world: `WorldInfo::build_world` (world.rs:64-76) on the GPU."""

    def __init__(self, info: WorldInfo, device: int = 0):
        self._lib = F.load()
        self._flat = info.flatten()
        err = C.create_string_buffer(512)
        self._handle = self._lib.mrt_scene_create(C.byref(self._flat.desc), device, err, len(err))
        if not self._handle:
            raise F.MrtError(F.MRT_ERR_INVALID_ARGUMENT, err.value.decode("utf-8", "replace"))
        self.device = device

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.mrt_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._handle

    def set_option(self, name: str, value: int) -> None:
        F.check(self._lib.mrt_scene_set_option(self._handle, name.encode(), int(value)))

    def bvh(self) -> np.ndarray:
        n = C.c_size_t(0)
        F.check(self._lib.mrt_scene_bvh(self._handle, None, 0, C.byref(n)))
        nodes = (F.BvhNode * max(1, n.value))()
        F.check(self._lib.mrt_scene_bvh(self._handle, nodes, n.value, C.byref(n)))
        return np.array([(list(x.bmin), list(x.bmax), x.object, x.skip) for x in nodes[:n.value]],
                        dtype=[("bmin", "f8", 3), ("bmax", "f8", 3), ("object", "i4"), ("skip", "i4")])

    def camera_rays(self, width: int, height: int, pixels: np.ndarray, seed: int, sample_index: int) -> np.ndarray:
        px = np.ascontiguousarray(pixels, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((len(px), 7), dtype=np.float64)
        F.check(self._lib.mrt_camera_rays(self._handle, width, height, px.ctypes.data, len(px), seed, sample_index,
                                          out.ctypes.data))
        return out

    def trace_rays(self, rays: np.ndarray, seed: int = 0):
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        n = len(r)
        obj = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float64)
        normal = np.zeros((n, 3), dtype=np.float64)
        pos = np.zeros((n, 3), dtype=np.float64)
        ff = np.zeros(n, dtype=np.int32)
        F.check(self._lib.mrt_trace_rays(self._handle, r.ctypes.data, n, seed, obj.ctypes.data, t.ctypes.data,
                                         normal.ctypes.data, pos.ctypes.data, ff.ctypes.data))
        return obj, t, normal, pos, ff


def _stats(s: F.Stats) -> RenderStats:
    return RenderStats(s.paths, s.rays, s.light_queries, s.launches, s.kernel_ms, s.total_ms, s.segment_kernel_avg_ms)


# CurrentShader::names() (ray_tracer.rs:310-318), indexed by MRT_SHADER_*
CURRENT_SHADER_NAMES = ["Ray Tracing", "Diffuse", "LightMap"]


class RayTracer:
    """`RayTracer` with `CurrentShader::Raytracing` (ray_tracer.rs:330-336)."""

    def __init__(self, info: WorldInfo, device: int = 0, seed: int = 0):
        self.world = World(info, device)
        self.seed = seed
        self._next_sample = 0
        self._lib = F.load()
        self.last_stats: Optional[RenderStats] = None

    def set_shader(self, shader) -> None:
        """`RayTracer::set_shader` (ray_tracer.rs:409-411). `shader`: a CurrentShader name
        ("Ray Tracing", "Diffuse", "LightMap"; ray_tracer.rs:310-329) or an MRT_SHADER_* value."""
        if isinstance(shader, str):
            if shader not in CURRENT_SHADER_NAMES:
                raise ValueError(f"invalid name: {shader}")
            shader = CURRENT_SHADER_NAMES.index(shader)
        self.world.set_option("shader", int(shader))

    def trace_part(self, part: ParallelImagePart, num_samples: int = 1, first_sample: Optional[int] = None) -> RenderStats:
        """`num_samples` x trace_part (ray_tracer.rs:435-463) over the strip, accumulated into part.image."""
        img = part.image
        fs = self._next_sample if first_sample is None else first_sample
        st = F.Stats()
        F.check(self._lib.mrt_render_accumulate(self.world.handle, img.buffer.ctypes.data, img.width, img.height,
                                                part.offset_x, part.width, num_samples, self.seed, fs, C.byref(st)))
        self.last_stats = _stats(st)
        return self.last_stats

    def trace_part_multi(self, others: list, part: ParallelImagePart, num_samples: int,
                         first_sample: Optional[int] = None) -> RenderStats:
        """trace_part with the samples split over this tracer's scene handle and those of `others`
        (normally one RayTracer per GPU built from the same WorldInfo), in one process."""
        img = part.image
        fs = self._next_sample if first_sample is None else first_sample
        handles = [self.world.handle] + [o.world.handle for o in others]
        arr = (C.c_void_p * len(handles))(*handles)
        st = F.Stats()
        F.check(self._lib.mrt_render_accumulate_multi(arr, len(handles), img.buffer.ctypes.data, img.width, img.height,
                                                      part.offset_x, part.width, num_samples, self.seed, fs, C.byref(st)))
        self.last_stats = _stats(st)
        return self.last_stats

    def trace_image(self, rgb_img: ParallelImage) -> RenderStats:  # ray_tracer.rs:413-418
        st = self.trace_part(rgb_img.split(1)[0], 1)
        self._next_sample += 1
        return st

    def post_process(self, rgb_img: ParallelImage, num_samples: int = 1) -> ParallelImage:
        """`RayTracer::post_process` (ray_tracer.rs:465-467): bloom (ray_tracer/bloom.rs:75-94) of
        rgb_img / num_samples, on the GPU. Images smaller than 128x128 raise (the reference panics)."""
        out = np.empty_like(rgb_img.buffer)
        F.check(self._lib.mrt_bloom(self.world.device, rgb_img.buffer.ctypes.data, rgb_img.width, rgb_img.height,
                                    num_samples, out.ctypes.data))
        return ParallelImage(out)

    def tracing_loop(self, parallel_image: ParallelImage, num_samples: int, post_process: bool = True) -> ParallelImage:
        """ray_tracer.rs:469-477: `num_samples` samples per pixel into `parallel_image`, then returns
        bloom(accumulation / num_samples). post_process=False returns the linear accumulation /
        num_samples instead (what the parity protocol compares)."""
        self.trace_part(parallel_image.split(1)[0], num_samples)
        self._next_sample += num_samples
        if post_process:
            return self.post_process(parallel_image, num_samples)
        return parallel_image / float(num_samples)

    def bloom_device(self, accum_ptr: int, width: int, height: int, num_samples: int, out_ptr: int, stream: int = 0) -> None:
        """bloom(accum / num_samples) between device-resident f32 buffers."""
        F.check(self._lib.mrt_bloom_device(self.world.device, accum_ptr, width, height, num_samples, out_ptr, stream))

    def to_image_device(self, rgb_ptr: int, width: int, height: int, num_samples: int, rgb8_ptr: int, stream: int = 0) -> None:
        """`ParallelImage::to_image` between device-resident buffers (uint8 output, top row first)."""
        F.check(self._lib.mrt_to_image_device(self.world.device, rgb_ptr, width, height, num_samples, rgb8_ptr, stream))

    def threaded_render(self, image: ParallelImage, samples_per_receive: int = 1) -> "ParallelImageCollector":
        """`RayTracer::threaded_render` (ray_tracer.rs:479-535): returns the collector that renders
        progressively into `image`."""
        return ParallelImageCollector(self, image, samples_per_receive)

    def render_device(self, accum_ptr: int, width: int, height: int, spp: int, first_sample: int = 0,
                      x0: int = 0, strip_w: Optional[int] = None, stream: int = 0) -> RenderStats:
        """Accumulates into a device-resident f32 buffer (e.g. a torch tensor's data_ptr())."""
        st = F.Stats()
        sw = width - x0 if strip_w is None else strip_w
        F.check(self._lib.mrt_render_accumulate_device(self.world.handle, accum_ptr, width, height, x0, sw, spp,
                                                       self.seed, first_sample, stream, C.byref(st)))
        self.last_stats = _stats(st)
        return self.last_stats


class ParallelImageCollector:
    """Progressive rendering with the interface of the reference's `ParallelImageCollector`
    (cpu_rt/src/parallel_image.rs:395-561) as returned by `RayTracer::threaded_render`
    (cpu_rt/src/ray_tracer.rs:479-535). The reference keeps eight worker threads that each add one sample
    per pixel to their strip and send it over a channel; `receive()` joins the newest strips, each
    divided by the number of samples it holds. Here one GPU renders the whole frame, so `receive()` adds
    `samples_per_receive` samples per pixel to one accumulation image and returns it divided by the
    samples taken so far (the same estimator, all strips always in step). `set_shader`, `clear` and
    `load_scenario` restart the accumulation as the reference's messages do (`part.set_black()`)."""

    def __init__(self, ray_tracer: "RayTracer", image: ParallelImage, samples_per_receive: int = 1):
        self.ray_tracer = ray_tracer
        self.image = image
        self.samples_per_receive = samples_per_receive
        self.num_samples = 0

    def clear(self) -> None:  # parallel_image.rs:415-417
        self.image.buffer[...] = 0.0
        self.num_samples = 0

    def receive(self) -> Optional[ParallelImage]:  # parallel_image.rs:418-438
        self.ray_tracer.trace_part(self.image.split(1)[0], self.samples_per_receive, first_sample=self.num_samples)
        self.num_samples += self.samples_per_receive
        return self.image / float(self.num_samples)

    def set_shader(self, shader) -> None:  # parallel_image.rs:477-487
        self.ray_tracer.set_shader(shader)
        self.clear()

    def load_scenario(self, info: WorldInfo) -> None:  # parallel_image.rs:439-462 (the caller builds the WorldInfo)
        device, seed = self.ray_tracer.world.device, self.ray_tracer.seed
        self.ray_tracer.world.close()
        self.ray_tracer.world = World(info, device)
        self.ray_tracer.seed = seed
        self.clear()

    def save_image(self, num_samples: int = 1) -> Optional[np.ndarray]:
        """The 8-bit image `save_file` writes (parallel_image.rs:463-476): the joined image, already a
        mean, mapped with num_samples = 1."""
        if self.num_samples == 0:
            return None
        return (self.image / float(self.num_samples)).to_image(num_samples)
