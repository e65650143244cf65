"""Generates tests/golden/hahog_berlin01.npz: a grey 640 x 480 version of the reference's own example image
(/root/reference/data/berlin/images/01.jpg, decoded and resized with PIL) and what the REFERENCE's features::hahog -- compiled from
/root/reference into oracle/_ref/libhahog_ref.so -- returns for it with OpenSfM's default thresholds (config.py:89-91:
hahog_peak_threshold 1e-5, hahog_edge_threshold 10) and 1500 features, post-processed as features.extract_features_hahog does
(square root, x 362, clip, round).  Run in the build container (needs /root/reference); the .npz travels with the repository."""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

im = Image.open("/root/reference/data/berlin/images/01.jpg").convert("L").resize((640, 480), Image.LANCZOS)
grey = np.asarray(im, np.uint8)
pts, desc = oracle.hahog_ref(grey.astype(np.float32) / 255, 1e-5, 10.0, 1500)
d8 = (362 * np.sqrt(desc)).clip(0, 255).round()
assert d8.max() <= 255 and np.array_equal(d8, d8.astype(np.uint8))
np.savez_compressed(os.path.join(HERE, "hahog_berlin01.npz"), grey=grey, points=pts, desc_u8=d8.astype(np.uint8), desc_f32_head=desc[:64])
print(grey.shape, pts.shape, desc.shape, "orientations per feature:", len(pts) / 1500.0)
