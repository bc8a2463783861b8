"""Pixels and ray counts that differ between the device render and the CPU checker at the same seed: flip_rate.py scene W H spp"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracing_amd import ParallelImage, RayTracer
from tests.helpers import scene_by_name
from tests.oracle_ffi import OracleWorld
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
info = scene_by_name(name)
rt = RayTracer(info, device=0, seed=1)
rt.world.set_option("trace_zero_throughput", 1)
img = ParallelImage.new_black(W, H)
st = rt.trace_part(img.split(1)[0], spp, first_sample=0)
ref, rst = OracleWorld(info).render(W, H, spp, seed=1, threads=16)
d = np.abs(ref - img.buffer).max(axis=2)
print(f"{name} {W}x{H}x{spp}: pixels differing by > 1e-3: {int((d > 1e-3).sum())} of {W * H} ({(d > 1e-3).mean() * 100:.4f} %), > 1e-6 relative to f32 rounding: {int((d > 1e-5 * (1 + np.abs(ref).max(axis=2))).sum())}; rays gpu {st.rays} checker {rst.rays}")
