"""MI355X path tracer behind the C ABI of include/mrt.h (replaces the cpu_rt per-pixel path-tracing loop)."""
from . import _ffi, scene, scenarios  # noqa: F401
from .ray_tracer import (ParallelImage, ParallelImageCollector, ParallelImagePart, RayTracer, RenderStats,  # noqa: F401
                         World)
