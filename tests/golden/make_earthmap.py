"""Decodes the reference's assets/earthmap.jpg into a committed data fixture (the image texture of the
"Light Demonstration" and "Easy Scene" scenarios): python tests/golden/make_earthmap.py  (needs /root/reference).

The reference decodes the file with the `image` crate (image 0.24.9 / jpeg-decoder 0.3.2,
cpu_rt/src/ray_tracer/texture.rs:196-215: `image::open(path).into_rgb8()`, texel = u8 / 255 as f32, row 0 = top
row of the file). This fixture is Pillow's (libjpeg-turbo) decode of the same bytes. Two baseline-JPEG decoders
may differ by one least-significant bit per channel (IDCT rounding, chroma up-sampling), so texel values are
"parity unpinned" within 1/255 against the reference's own decode; everything downstream of the texel fetch
is compared exactly as for every other scene.
"""
import hashlib
import os

import numpy as np
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/assets/earthmap.jpg"

raw = open(SRC, "rb").read()
im = Image.open(SRC).convert("RGB")
rgb8 = np.asarray(im, dtype=np.uint8)          # (height, width, 3), row 0 = top row of the file
np.savez_compressed(os.path.join(OUT, "earthmap_rgb8.npz"), rgb8=rgb8,
                    source_sha256=np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8))
print("earthmap", rgb8.shape, "mean", rgb8.mean(axis=(0, 1)), "sha256", hashlib.sha256(raw).hexdigest()[:16])
