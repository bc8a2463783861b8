"""A/B of option sets in one process (interleaved rounds). usage: perf_ab.py scene spp "opt=val,..." "opt=val,..." ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ray_tracing_amd import RayTracer
from tests.helpers import scene_by_name
name, spp = sys.argv[1], int(sys.argv[2])
cfgs = [eval("dict(" + a + ")") for a in sys.argv[3:]]
W, H = (3840, 2160) if name == "Voxel City" else (1920, 1080)
rt = RayTracer(scene_by_name(name), device=0, seed=0)
accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
defaults = dict(segments_per_launch=4096, pool_slots=1 << 21, traversal=0)
rt.render_device(accum.data_ptr(), W, H, 2)
for rep in range(3):
    for cfg in cfgs:
        for k, v in {**defaults, **cfg}.items():
            rt.world.set_option(k, v)
        accum.zero_()
        st = rt.render_device(accum.data_ptr(), W, H, spp)
        print(f"rep{rep} {cfg}: Mpaths/s {st.paths / st.total_ms / 1e3:8.1f} Mrays/s {st.rays / st.total_ms / 1e3:8.1f} "
              f"launches {st.launches} rays {st.rays} mean {accum.double().mean().item() / spp:.6f}", flush=True)
