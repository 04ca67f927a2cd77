#!/usr/bin/env python
"""Golden vectors for the uint8 Resize path (reference data_handling/data_class.py:61-71, inference.py:65-75:
``transforms.Resize(size)`` applied to a PIL image = ``Image.resize((w, h), Image.BILINEAR)``).  The arithmetic is the
third-party Pillow's (12.2.0 in this image); the vectors are Pillow outputs on seeded random images and on a crop of one of
the reference's training images, at the reference's own scale pairs divided down to fixture size.

    python tests/golden/make_golden_resize.py
"""
import os

import numpy as np
from PIL import Image
import PIL

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [((270, 480), (90, 160)),      # 2160 -> 720 (x 1/3), data_class.py:37
         ((216, 384), (108, 192)),     # 2160 -> 1080 (x 1/2)
         ((108, 192), (72, 128)),      # 1080 -> 720 (x 2/3)
         ((216, 384), (144, 256)),     # 2160 -> 1440
         ((135, 240), (6, 6)),         # 2160x3840 -> 96x96 geometry (aspect change), data_class.py:41
         ((96, 96), (64, 64)), ((64, 64), (96, 96)),
         ((37, 53), (20, 31)), ((20, 31), (37, 53)),
         ((50, 70), (50, 35)), ((50, 70), (25, 70)), ((33, 47), (33, 47))]


def main():
    rng = np.random.default_rng(2024)
    out = {"pillow_version": np.array(PIL.__version__)}
    for i, ((H, W), (h, w)) in enumerate(CASES):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        out[f"in_{i}"] = img
        out[f"size_{i}"] = np.array([h, w])
        out[f"out_{i}"] = np.asarray(Image.fromarray(img).resize((w, h), Image.BILINEAR))
    real = "/root/reference/images/training_set/image_9.png"
    if os.path.exists(real):
        im = Image.open(real).convert("RGB").crop((1400, 700, 1400 + 384, 700 + 216))
        out["real_in"] = np.asarray(im)
        out["real_out"] = np.asarray(im.resize((128, 72), Image.BILINEAR))
    out["n"] = np.array(len(CASES))
    np.savez_compressed(os.path.join(HERE, "resize_pil_cases.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
