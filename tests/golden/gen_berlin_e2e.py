"""Generates tests/golden/berlin_e2e.npz: what the end-to-end check of the reference's own example data set needs where /root/reference
does not exist (the GPU box).  Run in the build container.

* `gray_delta`: the three images of /root/reference/data/berlin/images as `features.extract_features` hands them to HAHOG
  (opensfm/features.py:594-611): decoded (PIL / libjpeg), reduced to feature_process_size 2048 the way `cv2.resize(INTER_AREA)` does it
  (cv2 is not installed here: the area weights of its `computeResizeAreaTab` are restated below), converted with cv2's fixed-point
  RGB2GRAY formula.  Image preparation is upstream of the hot path; it is test infrastructure here.
* `mask`: data/berlin/masks/*.png (first channel) at full resolution, for `masking.load_features_mask` (opensfm/masking.py:54-98).
* `ref_*`: the rows of data/berlin/tracks_example.csv (image, track id, feature id, x, y, scale) -- the only artefact in the reference that
  real HAHOG + cv2 matching + track creation produced.
"""
import collections
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/data/berlin"
NAMES = ["01.jpg", "02.jpg", "03.jpg"]
PROCESS_SIZE = 2048  # config.py: feature_process_size (the file predates data/berlin/config.yaml's 1024: its scales say 2048)


def area_tab(ssize: int, dsize: int) -> np.ndarray:
    """cv2 computeResizeAreaTab (imgproc/src/resize.cpp), as a dense dsize x ssize float32 weight matrix"""
    scale = ssize / dsize
    W = np.zeros((dsize, ssize), np.float32)
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            W[dx, sx1 - 1] = np.float32((sx1 - fsx1) / cell)
        W[dx, sx1:sx2] = np.float32(1.0 / cell)
        if fsx2 - sx2 > 1e-3:
            W[dx, sx2] = np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)
    return W


def resize_area(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    h, w = img.shape[:2]
    Wx, Wy = area_tab(w, dw).astype(np.float64), area_tab(h, dh).astype(np.float64)
    out = np.empty((dh, dw, img.shape[2]), np.uint8)
    for c in range(img.shape[2]):
        out[:, :, c] = np.clip(np.rint(Wy @ (img[:, :, c].astype(np.float64) @ Wx.T)), 0, 255).astype(np.uint8)
    return out


def rgb2gray(img: np.ndarray) -> np.ndarray:
    r, g, b = (img[:, :, i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)  # cv2 COLOR_RGB2GRAY, 14-bit fixed point


def main() -> None:
    grays, masks = [], []
    for n in NAMES:
        im = np.asarray(Image.open(f"{REF}/images/{n}").convert("RGB"))
        h, w = im.shape[:2]
        s = max(w, h)
        grays.append(rgb2gray(resize_area(im, w * PROCESS_SIZE // s, h * PROCESS_SIZE // s)))  # features.py:281-289
        m = np.asarray(Image.open(f"{REF}/masks/{n}.png"))
        masks.append(np.packbits((m[:, :, 0] if m.ndim == 3 else m) != 0, axis=1))
    rows = [line.rstrip("\n").split("\t") for line in open(f"{REF}/tracks_example.csv")][1:]
    gray = np.stack(grays)
    delta = gray.copy()  # stored as the difference to the left neighbour (mod 256): a quarter smaller after zlib, exact
    delta[:, :, 1:] = (gray[:, :, 1:].astype(np.int16) - gray[:, :, :-1]).astype(np.uint8)
    out = dict(gray_delta=delta, mask_bits=np.stack(masks), mask_shape=np.array(m.shape[:2]), full_size=np.array([w, h]),
               ref_image=np.array([NAMES.index(r[0]) for r in rows], np.int32), ref_track=np.array([int(r[1]) for r in rows], np.int32),
               ref_feature=np.array([int(r[2]) for r in rows], np.int32), ref_xys=np.array([[float(r[3]), float(r[4]), float(r[5])] for r in rows]))
    np.savez_compressed(os.path.join(HERE, "berlin_e2e.npz"), **out)
    print({k: v.shape for k, v in out.items()}, collections.Counter(out["ref_image"].tolist()))


if __name__ == "__main__":
    main()
