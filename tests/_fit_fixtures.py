"""Shared by tests/test_fit_oracle.py and tests/test_gpu_fit.py: readers of the demo-clip fixtures (tests/golden/demo_fit/README.md)."""
import glob
import json
import os

import numpy as np

ROOT = os.path.join(os.path.dirname(__file__), "golden", "demo_fit")
CLIPS = ["street", "indoor"]


def load_clip(name):
    from PIL import Image
    d = os.path.join(ROOT, name)
    z = np.load(os.path.join(d, "depth_intrinsics.npz"))
    depth = z["depth"].astype(np.float32)
    K = z["intrinsic"].astype(np.float32).copy()
    h, w = depth.shape
    K[0, 0] *= w; K[1, 1] *= h; K[0, 2] *= w; K[1, 2] *= h            # fit_3D_gaussian.py:508-512 (normalised intrinsics)
    masks = {int(os.path.basename(f).split("_")[1]): np.array(Image.open(f), dtype=np.uint8)
             for f in sorted(glob.glob(os.path.join(d, "masks", "mask_*.png")))}
    gold = json.load(open(os.path.join(d, "gaussian_params.json")))
    png = np.array(Image.open(os.path.join(d, "gaussian_projection.png")).convert("RGB"))
    return depth, K, masks, gold, png


def tab20():
    import matplotlib
    return [matplotlib.colormaps["tab20"](i)[:3] for i in range(20)]
