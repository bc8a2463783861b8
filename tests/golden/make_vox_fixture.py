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

# the tile map of the "Fast Oct Tree Sinnoh" scenario: the .yml's table plus every model it names
import yaml  # noqa: E402

m = yaml.safe_load(open("/root/reference/voxel_assets/sinnoh/twinleaf.yml"))
arrays = {"tiles": np.asarray(m["tiles"], dtype=np.int32), "tile_size": np.int32(m["tile_size"])}
types = []
for tt in m["tile_types"]:
    if tt.get("model_path"):
        vox = parse_vox(open(os.path.join("/root/reference", tt["model_path"]), "rb").read())
        k = int(tt["index"])
        types.append(k)
        arrays[f"m{k}_xyzi"] = np.concatenate(vox.models)
        arrays[f"m{k}_sizes"] = np.array([len(x) for x in vox.models], dtype=np.int64)
        arrays[f"m{k}_palette"] = vox.palette
arrays["model_types"] = np.asarray(types, dtype=np.int32)
np.savez_compressed(os.path.join(OUT, "sinnoh_map.npz"), **arrays)
print("sinnoh map", arrays["tiles"].shape, "tile size", int(arrays["tile_size"]), "models", types)
