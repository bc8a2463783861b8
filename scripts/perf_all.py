"""Throughput of every scenario at 1280x720 x 16 spp (one warm-up render first): perf_all.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import RayTracer, scenarios
from tests.helpers import scene_by_name
W, H, spp = 1280, 720, 16
names = list(scenarios.SCENARIOS) + ["Voxel City"]
for name in names:
    try:
        info = scene_by_name(name)
    except Exception as e:  # scenes that need a fixture this tree does not hold
        print(f"{name:36s} skipped: {e}")
        continue
    rt = RayTracer(info, device=0, seed=0)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device(accum.data_ptr(), W, H, 1)
    best, st = 0.0, None
    for rep in range(2):
        accum.zero_()
        st = rt.render_device(accum.data_ptr(), W, H, spp)
        best = max(best, st.paths / st.total_ms / 1e3)
    print(f"{name:36s} {best:8.1f} Mpaths/s  {st.rays / st.paths:5.2f} seg/path  objects {len(info.objects):4d}", flush=True)
