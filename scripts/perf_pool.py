"""Cornell smoke 1920x1080 at several path pool sizes: perf_pool.py spp slots..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracing_amd import RayTracer
from tests.helpers import scene_by_name
spp = int(sys.argv[1])
W, H = 1920, 1080
for slots in [int(a) for a in sys.argv[2:]]:
    rt = RayTracer(scene_by_name("Cornell Smoke"), device=0, seed=0)
    rt.world.set_option("pool_slots", slots)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device(accum.data_ptr(), W, H, 1)
    for rep in range(2):
        accum.zero_()
        st = rt.render_device(accum.data_ptr(), W, H, spp)
    print(f"slots {slots:8d}: {st.paths / st.total_ms / 1e3:8.1f} Mpaths/s  launches {st.launches}  kernel_ms {st.kernel_ms:.2f} total_ms {st.total_ms:.2f}", flush=True)
