"""Which traversal modes disagree with the automatic one on a generated test scene, and by how much."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ray_tracing_amd import RayTracer
from tests import oracle_ffi
from tests.helpers import probe_rays
from tests.test_random_scenes import random_scene
for seed in [int(a) for a in sys.argv[1:]] or [0]:
    info = random_scene(1000 + seed)
    ow = oracle_ffi.OracleWorld(info)
    rt = RayTracer(info, device=0, seed=seed)
    rays = probe_rays(ow, info, n_camera=10000, n_random=10000, seed=seed)
    rays[:, 6] = np.random.default_rng(seed).uniform(0.0, max(info.camera.end_time, 1e-9), len(rays))
    og, tg, ng, pg, fg = rt.world.trace_rays(rays, seed=seed)
    print("seed", seed, "objects", [type(o.shape).__name__ for o in info.objects])
    for mode in (1, 2, 3, 4):
        rt.world.set_option("traversal", mode)
        om, tm, _, _, _ = rt.world.trace_rays(rays, seed=seed)
        bad = np.nonzero((tm != tg) | (om != og))[0]
        print(" mode", mode, "differs on", len(bad), "rays; objects", np.unique(og[bad]), "max rel dt",
              (np.abs(tm[bad] - tg[bad]) / np.maximum(np.abs(tg[bad]), 1e-300)).max() if len(bad) else 0)
