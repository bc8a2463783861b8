"""MI355X path tracer behind the C ABI of include/mrt.h (replaces the cpu_rt per-pixel path-tracing loop)."""
from . import _ffi, scene, scenarios  # noqa: F401
from .ray_tracer import ParallelImage, ParallelImagePart, RayTracer, RenderStats, World  # noqa: F401
