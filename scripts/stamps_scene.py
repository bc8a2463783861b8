"""Region shares of the segment kernel on one scene (diagnostic build). usage: MRT_LIBRARY=.../libmrt_stamps.so stamps_scene.py scene W H spp"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import RayTracer
from tests.helpers import scene_by_name
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rt = RayTracer(scene_by_name(name), device=0, seed=0)
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    rt.world.set_option(k, int(v))
accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
st = rt.render_device(accum.data_ptr(), W, H, spp)
print(f"Mpaths/s {st.paths / st.total_ms / 1e3:.1f} Mrays/s {st.rays / st.total_ms / 1e3:.1f}")
