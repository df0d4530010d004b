"""Host-side mirror of the reference's utility/utils.py hot-path surface
(BoundBox :113, WeightReader :138, normalize :150, bbox_iou :155,
decode_netout :208, sigmoid :259, softmax :262, draw_boxes :190).

Same names, argument meaning and return shapes; the arithmetic of
decode_netout / NMS / bbox_iou runs on the MI355X through libmi355_dt.so
(mi355_dt.Context) -- there is no CPU implementation behind these calls.
"""
import numpy as np

import mi355_dt


class BoundBox(object):
    """utility/utils.py:113-136.  `track_id` is an addition (None unless set by
    MultiObjDetTracker)."""

    def __init__(self, x, y, w, h, c=None, classes=None):
        self.x = x
        self.y = y
        self.w = w
        self.h = h
        self.c = c
        self.classes = classes
        self.label = -1
        self.score = -1
        self.track_id = None

    def get_label(self):
        if self.label == -1:
            self.label = int(np.argmax(self.classes))
        return self.label

    def get_score(self):
        if self.score == -1:
            self.score = self.classes[self.get_label()]
        return self.score


class WeightReader(object):
    """utility/utils.py:138-148: flat float32 darknet file, reader offset starts
    at 4 floats (the 16-byte header)."""

    def __init__(self, weight_file):
        self.offset = 4
        self.all_weights = np.fromfile(weight_file, dtype="float32")

    def read_bytes(self, size):
        self.offset = self.offset + size
        return self.all_weights[self.offset - size:self.offset]

    def reset(self):
        self.offset = 4


def normalize(image):
    """utility/utils.py:150-153.  (On the device path the same x/255. is fused
    into conv_1's load; this host version exists for API parity.)"""
    return image / 255.


def sigmoid(x):
    return 1. / (1. + np.exp(-x))


def softmax(x, axis=-1, t=-100.):
    """utility/utils.py:262-270 -- including its global max / global rescale."""
    x = x - np.max(x)
    if np.min(x) < t:
        x = x / np.min(x) * t
    e_x = np.exp(x)
    return e_x / e_x.sum(axis, keepdims=True)


def _box4(b):
    return [b.x, b.y, b.w, b.h]


def bbox_iou(box1, box2):
    """utility/utils.py:155-173 for two BoundBox-like objects (device kernel)."""
    return float(bbox_iou_pairs(np.asarray([_box4(box1) + _box4(box2)], dtype=np.float32))[0])


def bbox_iou_pairs(pairs):
    """pairs [n,8] = (x,y,w,h) of box1 then box2 -> iou [n] (numpy)."""
    import torch
    ctx = mi355_dt.default_context()
    p = torch.as_tensor(np.ascontiguousarray(pairs, dtype=np.float32)).to(ctx.device)
    return ctx.bbox_iou(p).cpu().numpy()


def decode_netout(netout, obj_threshold, nms_threshold, anchors, nb_class):
    """utility/utils.py:208-257.  `netout` is a numpy array [GH,GW,NB,5+C]; like
    the reference it is transformed IN PLACE (conf, thresholded and NMS-zeroed
    class scores) and the returned boxes' `.classes` are views into it.
    Returns a list of BoundBox in the reference's (row,col,b) creation order."""
    boxes, _ = decode_netout_batch(netout[np.newaxis], obj_threshold, nms_threshold, anchors, nb_class,
                                   writeback=True)
    return boxes[0]


def decode_netout_batch(netouts, obj_threshold, nms_threshold, anchors, nb_class, writeback=False):
    """Batched decode: netouts numpy [B,GH,GW,NB,5+C] -> (list of B box lists, rows)
    where rows[b] is the raw [n,8] record array (x,y,w,h,conf,label,score,cell)."""
    import torch
    ctx = mi355_dt.default_context()
    src = netouts if isinstance(netouts, np.ndarray) else None
    arr = np.ascontiguousarray(netouts, dtype=np.float32)
    B, GH, GW, NB, S = arr.shape
    dev = torch.from_numpy(arr).to(ctx.device)
    r = ctx.decode(dev, obj_threshold, nms_threshold, anchors, nb_class, want_classes=not writeback,
                   want_post=writeback)
    counts = r["counts"].cpu().numpy()
    rows_all = r["boxes"].cpu().numpy()
    post = None
    if writeback:
        post = r["post"].cpu().numpy()
        if src is not None:
            src[...] = post          # the reference mutates its argument (utils.py:214-216,252), whatever its dtype / strides
            post = src
    classes = r["classes"].cpu().numpy() if r["classes"] is not None else None
    out, rows_out = [], []
    for b in range(B):
        n = int(counts[b])
        rows = rows_all[b, :n]
        lst = []
        for i in range(n):
            x, y, w, h, c, lab, sc, cell = rows[i]
            if post is not None:
                cell = int(cell)
                gb = cell % NB
                gc = (cell // NB) % GW
                gr = cell // (NB * GW)
                cls = post[b, gr, gc, gb, 5:]          # view, like the reference
            else:
                cls = classes[b, i]
            bb = BoundBox(x, y, w, h, c, cls)
            bb.label = int(lab)
            bb.score = sc
            lst.append(bb)
        out.append(lst)
        rows_out.append(rows)
    return out, rows_out


def generate_heatmap_feat(det_x, det_y, det_w, det_h, hmap_size=32):
    """utility/utils.py:53-58 (device kernel, float64 arguments as given)."""
    import torch
    ctx = mi355_dt.default_context()
    b = torch.as_tensor(np.asarray([[det_x, det_y, det_w, det_h]], dtype=np.float64)).to(ctx.device)
    return ctx.heatmap_from_xywh64(b, hmap_size)[0].cpu().numpy().astype(np.float64)


def generate_rectangle_from_heatmap(heat_map, thresh=0.75, hmap_size=32):
    """utility/utils.py:61-79 (device kernel): returns x1, y1, x2, y2."""
    import torch
    ctx = mi355_dt.default_context()
    h = torch.as_tensor(np.ascontiguousarray(heat_map, dtype=np.float32).reshape(1, -1)).to(ctx.device)
    x1, y1, x2, y2 = ctx.rect_from_heatmap(h, hmap_size, thresh)[0].cpu().numpy().tolist()
    return x1, y1, x2, y2


def draw_boxes(image, boxes, labels):
    """utility/utils.py:190-206 with PIL instead of OpenCV (cv2 is not part of
    this image).  `image` is an HxWx3 uint8 array; returns the annotated array."""
    from PIL import Image, ImageDraw
    im = Image.fromarray(np.ascontiguousarray(image[..., ::-1]))   # BGR (cv2 convention) -> RGB
    d = ImageDraw.Draw(im)
    H, W = image.shape[:2]
    for box in boxes:
        # w,h = anchor*exp(t)/G are unbounded; PIL (unlike cv2) rejects huge coordinates
        lim = lambda v, n: int(min(max(v, -4.0 * n), 5.0 * n))
        xmin = lim((box.x - box.w / 2) * W, W)
        xmax = lim((box.x + box.w / 2) * W, W)
        ymin = lim((box.y - box.h / 2) * H, H)
        ymax = lim((box.y + box.h / 2) * H, H)
        d.rectangle([xmin, ymin, max(xmax, xmin), max(ymax, ymin)], outline=(0, 255, 0), width=3)
        text = labels[box.get_label()] + ' ' + str(box.get_score())
        if getattr(box, "track_id", None) is not None:
            text = "#%d " % box.track_id + text
        d.text((xmin, max(0, ymin - 13)), text, fill=(0, 255, 0))
    return np.asarray(im)[..., ::-1].copy()
