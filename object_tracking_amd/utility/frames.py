"""Frame ingest for the predict entry points (host bookkeeping, not the hot path).

The reference does  cv2.imread -> cv2.resize(image,(IMAGE_H,IMAGE_W)) -> /255.
(models_detection/KerasYOLO.py:525-528).  OpenCV is not part of this image, so
decoding uses PIL; the resize runs on the device (csrc/ingest.hip: OpenCV's 8-bit
INTER_LINEAR scheme -- half-pixel centres, no anti-aliasing, 11-bit fixed-point
coefficients).  Parity with OpenCV itself is unpinned (cv2 absent).  Frames stay
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


def resize_bilinear_u8(image, out_h, out_w, ctx=None):
    """HxWx3 uint8 -> out_h x out_w x3 uint8 on the device (dt_ingest_resize: OpenCV's 8-bit
    INTER_LINEAR scheme, half-pixel centres, no anti-aliasing)."""
    import torch
    import mi355_dt
    ctx = ctx if ctx is not None else mi355_dt.default_context()
    d = torch.from_numpy(np.ascontiguousarray(image, dtype=np.uint8)[None]).to(ctx.device)
    return ctx.ingest_resize(d, out_h, out_w)[0].cpu().numpy()


def load_frame(path, image_h, image_w, ctx=None):
    """Returns (original BGR image, resized uint8 BGR frame)."""
    image = imread_bgr(path)
    return image, resize_bilinear_u8(image, image_h, image_w, ctx)
