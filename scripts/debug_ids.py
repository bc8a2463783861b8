import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracing_amd import scenarios, RayTracer
from tests.oracle_ffi import OracleWorld
from tests.helpers import probe_rays
info = scenarios.cornell_smoke()
ow = OracleWorld(info)
rt = RayTracer(info, device=0, seed=3)
rays = probe_rays(ow, info, n_camera=60000, n_random=60000, seed=21)
oo, to, no, po, fo = ow.trace_rays(rays, seed=5)
for mode in (1, 2):
    rt.world.set_option("traversal", mode)
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=5)
    bad = np.nonzero(og != oo)[0]
    print("mode", mode, "mismatches", len(bad), "camera-ray mismatches", int((bad < 60000).sum()))
    for i in bad[:12]:
        print("  ray", i, "oracle", oo[i], to[i], "gpu", og[i], tg[i], "o", rays[i, :3], "d", rays[i, 3:6])
    same = (og == oo) & (oo >= 0)
    print("  max rel t err on matching", (np.abs(tg[same] - to[same]) / np.abs(to[same])).max())
