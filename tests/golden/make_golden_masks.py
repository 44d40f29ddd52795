"""Generates tests/golden/masks.npz by IMPORTING the reference's own mask helpers (utils/wheatgs_utils.py: PILtoTorch,
binarize_mask, get_bbox_from_mask, calculate_seg_iou — the host-side scoring of run_3d_seg.py:88-89,127-163).
Run in the build container only (needs /root/reference):   python tests/golden/make_golden_masks.py
Only data (seeded inputs + the reference's outputs) is written."""
import importlib.util
import os

import numpy as np
import torch
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_wheatgs_utils", os.path.join(REF, "utils", "wheatgs_utils.py"))
U = importlib.util.module_from_spec(spec)
spec.loader.exec_module(U)

rng = np.random.RandomState(11)
out = {}
# (1) decoded 8-bit masks -> PILtoTorch -> binarize_mask (run_3d_seg.py:88-89)
for i, shape in enumerate(((75, 133), (48, 64, 3), (120, 160))):
    a = ((rng.rand(*shape) < 0.3) * rng.randint(1, 256, size=shape)).astype(np.uint8)
    if i == 2:
        a[a > 0] = 1                       # values of 1 / 255: still "inside" after the normalisation
    img = Image.fromarray(a)
    t = U.PILtoTorch(img, img.size)
    out[f"pixels{i}"] = a
    out[f"binary{i}"] = U.binarize_mask(t).squeeze(0).numpy()
# (2) alpha image -> alpha > 0.5 -> bbox, IoU against candidate masks (run_3d_seg.py:131-163)
H, W, K = 151, 259, 5
alpha = rng.rand(H, W).astype(np.float32)
alpha[:40] = 0.0
alpha[:, 250:] = 0.2
masks = rng.rand(K, H, W) < np.linspace(0.05, 0.9, K)[:, None, None]
masks[3] = False
pred = alpha > 0.5
out["alpha"], out["masks"] = alpha, masks
out["bbox"] = np.array(U.get_bbox_from_mask(pred.astype(np.float32)), np.int64)
out["iou"] = np.array([U.calculate_seg_iou(masks[k], pred) for k in range(K)], np.float64)
out["n_pred"] = np.int64(pred.sum())
out["bbox_empty_is_none"] = np.bool_(U.get_bbox_from_mask(np.zeros((4, 4), np.float32)) is None)
out["iou_empty_union"] = np.float64(U.calculate_seg_iou(np.zeros((4, 4), bool), np.zeros((4, 4), bool)))
np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
