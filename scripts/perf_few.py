"""Throughput of a few scenes with the library MRT_LIBRARY names: perf_few.py [scene:W:H:spp ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import RayTracer
from tests.helpers import scene_by_name
specs = sys.argv[1:] or ["Cornell Smoke:1920:1080:256", "Easy Cornell Box:1920:1080:64", "Random Scene:800:450:64"]
for spec in specs:
    name, W, H, spp = spec.split(":")
    W, H, spp = int(W), int(H), int(spp)
    rt = RayTracer(scene_by_name(name), device=0, seed=0)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device(accum.data_ptr(), W, H, 1)
    best = 0.0
    for rep in range(3):
        accum.zero_()
        st = rt.render_device(accum.data_ptr(), W, H, spp)
        best = max(best, st.paths / st.total_ms / 1e3)
    print(f"{name:24s} {W}x{H}x{spp}: best of 3 {best:8.1f} Mpaths/s", flush=True)
