"""Cuts a few 256 x 256 crops out of the reference's own test photos into tests/golden/real_crops.npz.

The reference trains and evaluates on crops of real photographs (/root/reference/licos/train.py:33-39,67-72:
``RandomCrop(256)`` / ``CenterCrop(256)`` + ``ToTensor()`` of the files under ``<root>/train``); the four JPEGs it
ships for its own smoke test (licos/tests/test_data/train/*.jpg) are the only real image data in this image.  This
script stores DATA only: uint8 pixel blocks (N, 3, 256, 256) - the centre crop of each photo (the reference's test
transform) plus seeded random crops at full resolution and at 1/4 scale (the box-averaged photo: a 256-crop then
spans 1024 source pixels, i.e. actual scene content instead of a flat patch of a 20-Mpixel frame).

    python tests/golden/make_real_crops.py          (needs /root/reference; run in the build container)
    python tests/golden/make_real_crops.py --pool   also writes build/real_pool.npz (git-ignored; 256 crops of 320 x 320
                                                    at scales 1, 1/2, 1/4, 1/8) - the training pool of
                                                    tools/train_weights.py --data real|mix
"""
import glob
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/licos/tests/test_data/train"
PER_IMAGE_FULL, PER_IMAGE_QUARTER = 3, 4


def main():
    rng = np.random.default_rng(2023)
    crops, names = [], []
    for path in sorted(glob.glob(os.path.join(SRC, "*.jpg"))):
        im = Image.open(path).convert("RGB")  # datasets/image.py ImageFolder: Image.open(p).convert("RGB")
        a = np.asarray(im)
        h, w = a.shape[:2]
        tag = os.path.basename(path)[-11:-4]
        y0, x0 = (h - 256) // 2, (w - 256) // 2
        crops.append(a[y0:y0 + 256, x0:x0 + 256])
        names.append(tag + ":center")
        for _ in range(PER_IMAGE_FULL):
            y0, x0 = int(rng.integers(0, h - 256)), int(rng.integers(0, w - 256))
            crops.append(a[y0:y0 + 256, x0:x0 + 256])
            names.append("%s:full@%d,%d" % (tag, y0, x0))
        q = np.asarray(im.resize((w // 4, h // 4), Image.BOX))
        for _ in range(PER_IMAGE_QUARTER):
            y0, x0 = int(rng.integers(0, q.shape[0] - 256)), int(rng.integers(0, q.shape[1] - 256))
            crops.append(q[y0:y0 + 256, x0:x0 + 256])
            names.append("%s:quarter@%d,%d" % (tag, y0, x0))
    x = np.ascontiguousarray(np.stack(crops).transpose(0, 3, 1, 2))
    out = os.path.join(HERE, "real_crops.npz")
    np.savez_compressed(out, x_u8=x, names=np.array(names))
    print(out, x.shape, x.dtype, os.path.getsize(out), "bytes; per-crop std",
          np.round(x.reshape(len(x), -1).std(1), 1).tolist())


def pool(n_per_scale=16, size=320):
    rng = np.random.default_rng(7)
    crops = []
    for path in sorted(glob.glob(os.path.join(SRC, "*.jpg"))):
        im = Image.open(path).convert("RGB")
        w, h = im.size
        for div in (1, 2, 4, 8):
            a = np.asarray(im if div == 1 else im.resize((w // div, h // div), Image.BOX))
            for _ in range(n_per_scale):
                y0, x0 = int(rng.integers(0, a.shape[0] - size)), int(rng.integers(0, a.shape[1] - size))
                crops.append(a[y0:y0 + size, x0:x0 + size])
    x = np.ascontiguousarray(np.stack(crops).transpose(0, 3, 1, 2))
    out = os.path.join(HERE, "..", "..", "build", "real_pool.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez(out, x_u8=x)
    print(os.path.abspath(out), x.shape, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
    if "--pool" in sys.argv:
        pool()
