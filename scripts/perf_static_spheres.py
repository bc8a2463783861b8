"""Throughput of the per-lane tree walk on a large static scene: Random Scene with every MovingSphere
replaced by a Sphere at its first centre (a probe, not a reference scenario)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import RayTracer, scenarios
from ray_tracing_amd.scene import MovingSphere, Object, Sphere
info = scenarios.random_scene()
info.objects = [Object(Sphere(o.shape.radius, o.shape.center_0, o.shape.material), o.transform)
                if isinstance(o.shape, MovingSphere) else o for o in info.objects]
W, H, spp = 1920, 1080, 16
for opts in [dict(traversal=1), dict(traversal=2)]:
    rt = RayTracer(info, device=0, seed=0)
    for k, v in opts.items():
        rt.world.set_option(k, v)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device(accum.data_ptr(), W, H, 1)
    accum.zero_()
    st = rt.render_device(accum.data_ptr(), W, H, spp)
    print(f"static spheres {opts}: Mpaths/s {st.paths / st.total_ms / 1e3:8.1f} Mrays/s {st.rays / st.total_ms / 1e3:8.1f} "
          f"seg/path {st.rays / st.paths:.2f} launches {st.launches} total_ms {st.total_ms:.1f}", flush=True)
