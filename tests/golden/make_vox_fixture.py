"""Decodes the reference's MagicaVoxel assets into committed data fixtures (inputs of the voxel-city
configuration): python tests/golden/make_vox_fixture.py  (needs /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ray_tracing_amd.octree import parse_vox  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
for name, path in [("skyscraper0", "voxel_assets/city/skyscraper0.vox"), ("vox_2x2", "voxel_assets/2x2.vox"),
                   ("apartment_building", "voxel_assets/apartment_building.vox")]:
    vox = parse_vox(open(os.path.join("/root/reference", path), "rb").read())
    sizes = np.array([len(m) for m in vox.models], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, name + "_vox.npz"), xyzi=np.concatenate(vox.models), model_sizes=sizes,
                        palette=vox.palette)
    allv = np.concatenate(vox.models)
    print(name, "models", len(vox.models), "voxels", len(allv), "extent", allv[:, :3].min(0), allv[:, :3].max(0),
          "colours used", len(np.unique(allv[:, 3])))
