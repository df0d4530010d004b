"""Frame ingest for the predict entry points (host bookkeeping, not the hot path).

The reference does  cv2.imread -> cv2.resize(image,(IMAGE_H,IMAGE_W)) -> /255.
(models_detection/KerasYOLO.py:525-528).  OpenCV is not part of this image, so
decoding uses PIL and the resize is this file's own bilinear with half-pixel
centres and no anti-aliasing (cv2.INTER_LINEAR's definition).  Parity with
OpenCV itself is unpinned (cv2 absent).  Frames stay
uint8 BGR (cv2.imread's channel order; the reference's predict path never flips
to RGB); the /255. happens on the device, fused into conv_1.
"""
import numpy as np


def imread_bgr(path):
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


def imwrite_bgr(path, image):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(image[..., ::-1])).save(path)


def resize_bilinear_u8(image, out_h, out_w):
    """HxWx3 uint8 -> out_h x out_w x3 uint8, half-pixel-centre bilinear."""
    H, W = image.shape[:2]
    if (H, W) == (out_h, out_w):
        return np.ascontiguousarray(image)
    ys = (np.arange(out_h, dtype=np.float64) + 0.5) * (H / float(out_h)) - 0.5
    xs = (np.arange(out_w, dtype=np.float64) + 0.5) * (W / float(out_w)) - 0.5
    y0 = np.floor(ys).astype(np.int64); x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    y0c = np.clip(y0, 0, H - 1); y1c = np.clip(y0 + 1, 0, H - 1)
    x0c = np.clip(x0, 0, W - 1); x1c = np.clip(x0 + 1, 0, W - 1)
    im = image.astype(np.float64)
    top = im[y0c][:, x0c] * (1 - fx) + im[y0c][:, x1c] * fx
    bot = im[y1c][:, x0c] * (1 - fx) + im[y1c][:, x1c] * fx
    out = top * (1 - fy) + bot * fy
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def load_frame(path, image_h, image_w):
    """Returns (original BGR image, resized uint8 BGR frame)."""
    image = imread_bgr(path)
    return image, resize_bilinear_u8(image, image_h, image_w)
