#!/usr/bin/env python
"""Generate tests/golden/clip.npz by IMPORTING the reference's ClipToTensor + Normalize (build container only).

    python tools/make_golden_clip.py [--ref /root/reference] [--out tests/golden]

Stores the uint8 input frames and the reference's float output; no reference source is copied.
"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    from utils.videotransforms import volume_transforms, tensor_transforms   # reference
    rng = np.random.RandomState(20260928)
    out = {}
    for tag, (B, T, H, W) in {"w8": (2, 4, 6, 8), "w7": (2, 3, 5, 7)}.items():
        frames = rng.randint(0, 256, (B, T, H, W, 3)).astype(np.uint8)
        frames[0, 0, 0, :2] = [[0, 0, 0], [255, 255, 255]]                 # the extremes
        ref = []
        for b in range(B):
            t = volume_transforms.ClipToTensor()([frames[b, i] for i in range(T)])
            t = tensor_transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(t)
            ref.append(t.numpy())
        out[f"{tag}_frames"] = frames
        out[f"{tag}_out"] = np.stack(ref).astype(np.float32)
    np.savez_compressed(os.path.join(args.out, "clip.npz"), **out)
    print("clip.npz", os.path.getsize(os.path.join(args.out, "clip.npz")))


if __name__ == "__main__":
    main()
