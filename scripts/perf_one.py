"""Throughput of one scene: perf_one.py scene W H spp [opt=val ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import RayTracer
from tests.helpers import scene_by_name
name, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rt = RayTracer(scene_by_name(name), device=0, seed=0)
for a in sys.argv[5:]:
    k, v = a.split("=")
    rt.world.set_option(k, int(v))
accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
rt.render_device(accum.data_ptr(), W, H, 1)
for rep in range(2):
    accum.zero_()
    st = rt.render_device(accum.data_ptr(), W, H, spp)
    print(f"{name} {W}x{H}x{spp} {sys.argv[5:]}: Mpaths/s {st.paths / st.total_ms / 1e3:8.1f} Mrays/s {st.rays / st.total_ms / 1e3:8.1f} "
          f"seg/path {st.rays / st.paths:.2f} launches {st.launches} total_ms {st.total_ms:.1f} mean {accum.double().mean().item() / spp:.6f}", flush=True)
