"""Candidate search against the tree walk on one scene: identical nearest hits, then render rates.
usage: cand_probe.py [scene] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ray_tracing_amd import RayTracer
from tests.helpers import scene_by_name
name = sys.argv[1] if len(sys.argv) > 1 else "Random Scene"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
info = scene_by_name(name)
rt = RayTracer(info, device=0, seed=0)
W, H = 800, 450
rng = np.random.default_rng(1)
px = np.stack([rng.integers(0, W, 200000), rng.integers(0, H, 200000)], axis=1).astype(np.int32)
rays = rt.world.camera_rays(W, H, px, 5, 0)
if rays is not None:
    # second-generation rays: from the hit points into random directions
    rt.world.set_option("traversal", 1)
    o1, t1, n1, p1, f1 = rt.world.trace_rays(rays, seed=9)
    sec = rays.copy()
    hit = o1 >= 0
    sec[hit, 0:3] = p1[hit] + 1e-3 * n1[hit]
    sec[:, 3:6] = rng.normal(size=(len(rays), 3))
    both = np.concatenate([rays, sec])
    rt.world.set_option("traversal", 1)
    oa, ta, na, pa, fa = rt.world.trace_rays(both, seed=9)
    rt.world.set_option("traversal", 4)
    ob, tb, nb, pb, fb = rt.world.trace_rays(both, seed=9)
    print("rays", len(both), "hits", int((oa >= 0).sum()), "id mismatches", int((oa != ob).sum()), "t mismatches", int((ta != tb).sum()), flush=True)
accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
for mode in (1, 4, 1, 4):
    rt.world.set_option("traversal", mode)
    accum.zero_()
    st = rt.render_device(accum.data_ptr(), W, H, spp)
    print(f"traversal={mode}: Mpaths/s {st.paths / st.total_ms / 1e3:8.1f} Mrays/s {st.rays / st.total_ms / 1e3:8.1f} rays {st.rays} "
          f"mean {accum.double().mean().item() / spp:.7f}", flush=True)
