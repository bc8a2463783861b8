"""Throughput probe: Cornell smoke at several option settings (one process, interleaved)."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests.helpers import scene_by_name
from ray_tracing_amd import scenarios, RayTracer

def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Cornell Smoke"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    W, H = 1920, 1080
    info = scene_by_name(name)
    rt = RayTracer(info, device=0, seed=0)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    configs = [dict(segments_per_launch=s, pool_slots=p) for s in (2, 4, 8, 16) for p in (1 << 21, 1 << 22, 1 << 23)]
    if len(sys.argv) > 3:
        configs = [eval("dict(" + sys.argv[3] + ")")]
    rt.render_device(accum.data_ptr(), W, H, 4)
    for rep in range(2):
        for cfg in configs:
            for k, v in cfg.items():
                rt.world.set_option(k, v)
            accum.zero_()
            st = rt.render_device(accum.data_ptr(), W, H, spp)
            print(f"rep{rep} {cfg}: Mpaths/s {st.paths / st.total_ms / 1e3:8.1f} Mrays/s {st.rays / st.total_ms / 1e3:8.1f} "
                  f"launches {st.launches} kernel_ms {st.kernel_ms:.1f} total_ms {st.total_ms:.1f} mean {accum.mean().item() / spp:.5f}", flush=True)
main()
