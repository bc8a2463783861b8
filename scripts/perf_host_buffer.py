"""PCIe-inclusive rate: the host-buffer entry point against the device-resident one, same work."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ray_tracing_amd import ParallelImage, RayTracer, scenarios
W, H = 1920, 1080
rt = RayTracer(scenarios.cornell_smoke(), device=0, seed=0)
accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
rt.render_device(accum.data_ptr(), W, H, 4)
img = ParallelImage.new_black(W, H)
rt.trace_part(img.split(1)[0], 4, first_sample=0)
for spp in (64, 1024):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = rt.render_device(accum.data_ptr(), W, H, spp)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st2 = rt.trace_part(img.split(1)[0], spp, first_sample=0)
    t2 = time.perf_counter()
    print(f"spp {spp}: device buffer {st.paths / (t1 - t0) / 1e6:8.1f} Mpaths/s ({(t1 - t0) * 1e3:.1f} ms)  "
          f"host buffer {st2.paths / (t2 - t1) / 1e6:8.1f} Mpaths/s ({(t2 - t1) * 1e3:.1f} ms)", flush=True)
