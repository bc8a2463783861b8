import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.helpers import scene_by_name
from ray_tracing_amd import scenarios, RayTracer
info = scene_by_name(sys.argv[1] if len(sys.argv) > 1 else "Cornell Smoke")
rt = RayTracer(info, device=0, seed=3)
W, H = 1920, 1080
ys, xs = np.mgrid[0:H, 0:W]
px = np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)
cam = rt.world.camera_rays(W, H, px, 1, 0)
for mode in (1, 2):
    rt.world.set_option("traversal", mode)
    obj, t, n, p, ff = rt.world.trace_rays(cam, seed=1)
    # secondary-like rays: from hit points, random directions
    rng = np.random.default_rng(0)
    hit = obj >= 0
    sec = cam.copy()
    sec[hit, 0:3] = p[hit] + 1e-3 * n[hit]
    d = rng.normal(size=(len(cam), 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec[:, 3:6] = d
    rt.world.trace_rays(sec, seed=2)
print("done")
