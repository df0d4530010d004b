"""ctypes binding of libmi355_dt.so (include/mi355_dt.h) for the Python host.

PyTorch-ROCm is used only as plumbing: device buffers (`tensor.data_ptr()`),
the current HIP stream and `torch.distributed`.  All compute goes through the
C ABI.  There is NO CPU fallback: if the shared library is missing, or no gfx950
device is visible, this module raises -- it never routes around the HIP path.

The binding style follows the reference's own ctypes precedent,
models_detection/YOLO.py:6-37,58-119 (libdarknet.so), with status codes added.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_DT_LIB") or os.path.join(_HERE, "libmi355_dt.so")   # override: ablation builds only

DT_FRAMES_U8 = 0
DT_FRAMES_F32 = 1
DT_BOX_FLOATS = 8

SYMBOLS = [
    "dt_create", "dt_destroy", "dt_last_error", "dt_set_stream", "dt_abi_version",
    "dt_detector_config", "dt_load_darknet_weights", "dt_detect_forward", "dt_detector_tap", "dt_ingest_resize",
    "dt_decode", "dt_bbox_iou", "dt_tracker_load", "dt_track_forward", "dt_associate",
    "dt_tiny_load", "dt_tiny_forward", "dt_tiny_features", "dt_tiny_sequence", "dt_top_box", "dt_heatmap_from_boxes", "dt_heatmap_from_xywh64", "dt_rect_from_heatmap", "dt_encode_targets", "dt_graph_enable", "dt_conv2d", "dt_convlstm_step",
    "dt_profile_enable", "dt_profile_reset", "dt_profile_read", "dt_profile_names", "dt_policy_reload", "dt_detector_extract", "dt_decode_per_frame",
    "dt_track_row_width", "dt_track_detect", "dt_track_recurrent",
    "dt_packed_row_ints", "dt_pack_detections", "dt_unpack_detections",
    "dt_track_xproj_width", "dt_track_detect_xproj", "dt_track_recurrent_xproj",
    "dt_gemm_split_bf16", "dt_gemm_split", "dt_policy_set",
]

_lib = None


class NativeError(RuntimeError):
    pass


def load_library():
    """dlopen libmi355_dt.so and declare the prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "libmi355_dt.so not found at %s -- build it with `python -m object_tracking_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf, csz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.dt_create.argtypes = [ctypes.POINTER(vp)]
    L.dt_destroy.argtypes = [vp]
    L.dt_destroy.restype = None
    L.dt_last_error.argtypes = [vp]
    L.dt_last_error.restype = ctypes.c_char_p
    L.dt_set_stream.argtypes = [vp, vp]
    L.dt_abi_version.argtypes = []
    L.dt_detector_config.argtypes = [vp, ci, ci, ci, ci, vp]
    L.dt_load_darknet_weights.argtypes = [vp, vp, csz, ctypes.POINTER(csz)]
    L.dt_detect_forward.argtypes = [vp, vp, ci, ci, vp, vp]
    L.dt_detector_tap.argtypes = [vp, ctypes.c_char_p, ci, vp]
    L.dt_detector_extract.argtypes = [vp, vp, ci, ci, ctypes.c_char_p, vp, csz, ctypes.POINTER(ci)]
    L.dt_ingest_resize.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci]
    L.dt_decode.argtypes = [vp, vp, ci, ci, ci, ci, ci, cf, cf, vp, ci, vp, vp, vp, vp]
    L.dt_bbox_iou.argtypes = [vp, vp, ci, vp]
    L.dt_decode_per_frame.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp]
    L.dt_tracker_load.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.dt_track_forward.argtypes = [vp, vp, ci, ci, ci, vp, vp]
    L.dt_associate.argtypes = [vp, vp, vp, ci, ci, ci, cf, vp, vp]
    L.dt_track_row_width.argtypes = [vp]
    L.dt_track_detect.argtypes = [vp, vp, ci, ci, vp]
    L.dt_track_recurrent.argtypes = [vp, vp, ci, ci, vp, vp]
    L.dt_track_xproj_width.argtypes = [vp]
    L.dt_track_detect_xproj.argtypes = [vp, vp, ci, ci, vp, vp]
    L.dt_track_recurrent_xproj.argtypes = [vp, vp, ci, ci, vp]
    L.dt_packed_row_ints.argtypes = [ci, ci]
    L.dt_pack_detections.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    L.dt_unpack_detections.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    L.dt_tiny_load.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp]
    L.dt_heatmap_from_boxes.argtypes = [vp, vp, ci, ci, vp]
    L.dt_heatmap_from_xywh64.argtypes = [vp, vp, ci, ci, vp]
    L.dt_rect_from_heatmap.argtypes = [vp, vp, ci, ci, cf, vp]
    L.dt_encode_targets.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp]
    L.dt_tiny_forward.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.dt_tiny_features.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.dt_tiny_sequence.argtypes = [vp, vp, ci, ci, vp]
    L.dt_top_box.argtypes = [vp, vp, vp, ci, ci, vp]
    L.dt_conv2d.argtypes = [vp, vp, ci, ci, ci, ci, vp, ci, ci, vp, cf, ci, vp, vp]
    L.dt_convlstm_step.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp]
    L.dt_gemm_split_bf16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    L.dt_gemm_split.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.dt_profile_enable.argtypes = [vp, ci]
    L.dt_graph_enable.argtypes = [vp, ci]
    L.dt_profile_reset.argtypes = [vp]
    L.dt_policy_reload.argtypes = [vp]
    L.dt_policy_set.argtypes = [vp, ctypes.c_char_p, ci]
    L.dt_profile_names.argtypes = [vp, ctypes.c_char_p, csz]
    L.dt_profile_read.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64),
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                  ctypes.POINTER(ctypes.c_double)]
    for s in SYMBOLS:
        if s not in ("dt_destroy", "dt_last_error", "dt_packed_row_ints"):
            getattr(L, s).restype = ctypes.c_int
    L.dt_packed_row_ints.restype = ctypes.c_size_t
    _lib = L
    return L


def _hptr(a):
    """host float32 array -> (keepalive, void*)"""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _dptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Context(object):
    """One dt_ctx (one GPU, one process)."""

    def __init__(self, device=None):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise NativeError("no HIP device visible to PyTorch-ROCm; the MI355X path has no CPU fallback")
        if device is not None:
            torch.cuda.set_device(device)
        self.device = torch.device("cuda", torch.cuda.current_device())
        h = ctypes.c_void_p()
        rc = self.lib.dt_create(ctypes.byref(h))
        if rc != 0:
            raise NativeError("dt_create failed (%d): %s" % (rc, self.lib.dt_last_error(None).decode()))
        self.h = h
        self.cb = None
        self.grid = None
        self.nb_box = None
        self.nb_class = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.dt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise NativeError("%s failed (%d): %s" % (what, rc, self.lib.dt_last_error(self.h).decode()))

    def _sync_stream(self):
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        self.lib.dt_set_stream(self.h, ctypes.c_void_p(st))

    def _f32(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.device)

    # ---- detector -----------------------------------------------------
    def detector_config(self, image_h, image_w, nb_box, nb_class, anchors):
        keep, p = _hptr(anchors)
        self._check(self.lib.dt_detector_config(self.h, image_h, image_w, nb_box, nb_class, p), "dt_detector_config")
        self.image_h, self.image_w = image_h, image_w
        self.nb_box, self.nb_class = nb_box, nb_class
        self.cb = nb_box * (5 + nb_class)
        self.grid = (image_h // 32, image_w // 32)

    def load_darknet_weights(self, blob):
        keep, p = _hptr(blob)
        used = ctypes.c_size_t(0)
        self._check(self.lib.dt_load_darknet_weights(self.h, p, keep.size, ctypes.byref(used)), "dt_load_darknet_weights")
        return used.value

    def _frames_dtype(self, frames):
        t = self.torch
        if frames.dtype == t.uint8:
            return DT_FRAMES_U8
        if frames.dtype == t.float32:
            return DT_FRAMES_F32
        raise NativeError("frames must be uint8 or float32, got %s" % frames.dtype)

    def detect_forward(self, frames, want_feat=False):
        """frames [B,H,W,3] device tensor -> netout [B,G,G,nb_box,5+C] (+ feat [B,G,G,1024])."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dim() == 4
        B = frames.shape[0]
        gh, gw = self.grid
        netout = self._f32(B, gh, gw, self.nb_box, 5 + self.nb_class)
        feat = self._f32(B, gh, gw, 1024) if want_feat else None
        self._sync_stream()
        self._check(self.lib.dt_detect_forward(self.h, _dptr(frames), self._frames_dtype(frames), B,
                                               _dptr(netout), _dptr(feat)), "dt_detect_forward")
        return (netout, feat) if want_feat else netout

    def detect_forward_internal(self, frames):
        """Forward into the context's own workspaces (for dt_detector_tap)."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dim() == 4
        self._sync_stream()
        self._check(self.lib.dt_detect_forward(self.h, _dptr(frames), self._frames_dtype(frames), frames.shape[0],
                                               None, None), "dt_detect_forward")

    def detector_tap(self, name, batch):
        H, W = self.image_h, self.image_w
        shape = {"act_13": (batch, H // 16, W // 16, 512), "conv_feat": (batch, H // 32, W // 32, 1024),
                 "conv_23": (batch, H // 32, W // 32, self.cb)}[name]
        out = self._f32(*shape)
        self._sync_stream()
        self._check(self.lib.dt_detector_tap(self.h, name.encode(), batch, _dptr(out)), "dt_detector_tap")
        return out

    def layer_shape(self, layer, batch=1):
        """(batch, h, w, channels) of a named detector layer; raises NativeError for an unknown name."""
        shape = (ctypes.c_int * 4)()
        self._check(self.lib.dt_detector_extract(self.h, None, 0, batch, layer.encode(), None, 0, shape), "dt_detector_extract")
        return tuple(int(v) for v in shape)

    def detector_extract(self, frames, layer):
        """frames [B,H,W,3] device tensor -> the named layer's output [B,h,w,C] (KerasYOLO.extract, any layer)."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dim() == 4
        B = frames.shape[0]
        shape = (ctypes.c_int * 4)()
        self._check(self.lib.dt_detector_extract(self.h, None, 0, B, layer.encode(), None, 0, shape), "dt_detector_extract")
        out = self._f32(*[int(v) for v in shape])
        self._sync_stream()
        self._check(self.lib.dt_detector_extract(self.h, _dptr(frames), self._frames_dtype(frames), B, layer.encode(),
                                                 _dptr(out), out.numel(), shape), "dt_detector_extract")
        return out

    # ---- frame ingest -------------------------------------------------
    def ingest_resize(self, frames, out_h, out_w):
        """frames uint8 [n,Hs,Ws,3] device tensor -> uint8 [n,out_h,out_w,3] (cv2.resize INTER_LINEAR)."""
        t = self.torch
        assert frames.is_cuda and frames.dtype == t.uint8 and frames.is_contiguous() and frames.dim() == 4
        n, Hs, Ws, _ = frames.shape
        out = t.empty((n, out_h, out_w, 3), dtype=t.uint8, device=self.device)
        self._sync_stream()
        self._check(self.lib.dt_ingest_resize(self.h, _dptr(frames), n, Hs, Ws, _dptr(out), out_h, out_w),
                    "dt_ingest_resize")
        return out

    # ---- decode -------------------------------------------------------
    def decode(self, netout, obj_threshold, nms_threshold, anchors, nb_class, cap=None,
               want_classes=False, want_post=False):
        """netout [B,GH,GW,NB,5+C] device float32 (not modified).  Returns dict of
        device tensors: boxes [B,cap,8], counts [B] (+ classes, post).  obj_threshold / nms_threshold are
        scalars, or arrays / tensors of one value per frame (dt_decode_per_frame)."""
        t = self.torch
        assert netout.is_cuda and netout.dtype == t.float32 and netout.is_contiguous() and netout.dim() == 5
        B, GH, GW, NB, S = netout.shape
        assert S == 5 + nb_class
        if cap is None:
            cap = GH * GW * NB
        boxes = t.empty((B, cap, DT_BOX_FLOATS), dtype=t.float32, device=self.device)   # the kernel zero-fills rows >= count
        counts = t.empty((B,), dtype=t.int32, device=self.device)
        classes = t.zeros((B, cap, nb_class), dtype=t.float32, device=self.device) if want_classes else None
        post = t.empty_like(netout) if want_post else None
        keep, ap = _hptr(anchors)
        self._sync_stream()
        if np.ndim(obj_threshold) or np.ndim(nms_threshold) or t.is_tensor(obj_threshold) or t.is_tensor(nms_threshold):
            def per_frame(v):
                v = v.detach().cpu().numpy() if t.is_tensor(v) else np.asarray(v)
                return np.broadcast_to(v.astype(np.float32).reshape(-1), (B,))
            thr = t.from_numpy(np.ascontiguousarray(np.stack([per_frame(obj_threshold), per_frame(nms_threshold)], 1))).to(self.device)
            self._check(self.lib.dt_decode_per_frame(self.h, _dptr(netout), B, GH, GW, NB, nb_class, _dptr(thr), ap, cap,
                                                     _dptr(boxes), _dptr(counts), _dptr(classes), _dptr(post)),
                        "dt_decode_per_frame")
        else:
            self._check(self.lib.dt_decode(self.h, _dptr(netout), B, GH, GW, NB, nb_class, float(obj_threshold),
                                           float(nms_threshold), ap, cap, _dptr(boxes), _dptr(counts),
                                           _dptr(classes), _dptr(post)), "dt_decode")
        return dict(boxes=boxes, counts=counts, classes=classes, post=post)

    def bbox_iou(self, pairs):
        t = self.torch
        assert pairs.is_cuda and pairs.dtype == t.float32 and pairs.is_contiguous()
        n = pairs.shape[0]
        out = self._f32(n)
        self._sync_stream()
        self._check(self.lib.dt_bbox_iou(self.h, _dptr(pairs), n, _dptr(out)), "dt_bbox_iou")
        return out

    def associate(self, boxes, counts, assoc_threshold):
        """boxes [n_clips,T,cap,8], counts [n_clips,T] int32 -> ids [n_clips,T,cap], nids [n_clips]."""
        t = self.torch
        assert boxes.is_cuda and boxes.is_contiguous() and counts.is_contiguous() and counts.dtype == t.int32
        n_clips, T, cap, _ = boxes.shape
        ids = t.empty((n_clips, T, cap), dtype=t.int32, device=self.device)
        nids = t.empty((n_clips,), dtype=t.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.dt_associate(self.h, _dptr(boxes), _dptr(counts), n_clips, T, cap,
                                          float(assoc_threshold), _dptr(ids), _dptr(nids)), "dt_associate")
        return ids, nids

    # ---- cross-stream exchange ------------------------------------------
    def pack_detections(self, boxes, counts, ids, nids, n_rows):
        """this rank's detection table -> int32 rows [n_rows, dt_packed_row_ints(T, cap)] (rows past the local clips
        are empty): the buffer of the ONE all-gather of a step."""
        t = self.torch
        n, T, cap = ids.shape
        assert boxes.is_cuda and boxes.is_contiguous() and counts.is_contiguous() and ids.is_contiguous() and nids.is_contiguous()
        assert counts.dtype == t.int32 and ids.dtype == t.int32 and nids.dtype == t.int32 and boxes.dtype == t.float32
        assert n <= n_rows, "this rank holds %d clips but the exchange was sized for at most %d per rank (n_clips_max)" % (n, n_rows)
        rows = t.empty((n_rows, int(self.lib.dt_packed_row_ints(T, cap))), dtype=t.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.dt_pack_detections(self.h, _dptr(boxes), _dptr(counts), _dptr(ids), _dptr(nids), n, T, cap,
                                                n_rows, _dptr(rows)), "dt_pack_detections")
        return rows

    def unpack_detections(self, rows, T, cap):
        """gathered rows of all ranks [R, row] -> (boxes, counts, ids, nids, gids) of the valid clips in global order"""
        t = self.torch
        assert rows.is_cuda and rows.is_contiguous() and rows.dtype == t.int32
        R = rows.shape[0]
        assert rows.shape[1] == int(self.lib.dt_packed_row_ints(T, cap))
        boxes = t.empty((R, T, cap, DT_BOX_FLOATS), dtype=t.float32, device=self.device)
        counts = t.empty((R, T), dtype=t.int32, device=self.device)
        ids = t.empty((R, T, cap), dtype=t.int32, device=self.device)
        nids = t.empty((R,), dtype=t.int32, device=self.device)
        gids = t.empty((R, T, cap), dtype=t.int64, device=self.device)
        nv = t.zeros((1,), dtype=t.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.dt_unpack_detections(self.h, _dptr(rows), R, T, cap, _dptr(boxes), _dptr(counts), _dptr(ids),
                                                  _dptr(nids), _dptr(gids), _dptr(nv)), "dt_unpack_detections")
        n = int(nv.item())
        return boxes[:n], counts[:n], ids[:n], nids[:n], gids[:n]

    # ---- tracker ------------------------------------------------------
    def tracker_load(self, units, kernel, recurrent, bias, out_kernel, out_bias):
        ks = [_hptr(a) for a in (kernel, recurrent, bias, out_kernel, out_bias)]
        self._check(self.lib.dt_tracker_load(self.h, units, *[k[1] for k in ks]), "dt_tracker_load")
        self.trk_units = units

    def track_forward(self, frames, want_det=True):
        """frames [n_clips,T,H,W,3] -> tracking grid [n_clips,T,G,G,NB,5+C] (+ detection grid)."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dim() == 5
        n_clips, T = frames.shape[:2]
        gh, gw = self.grid
        trk = self._f32(n_clips, T, gh, gw, self.nb_box, 5 + self.nb_class)
        det = self._f32(n_clips, T, gh, gw, self.nb_box, 5 + self.nb_class) if want_det else None
        self._sync_stream()
        self._check(self.lib.dt_track_forward(self.h, _dptr(frames), self._frames_dtype(frames), n_clips, T,
                                              _dptr(trk), _dptr(det)), "dt_track_forward")
        return (trk, det) if want_det else trk

    def track_row_width(self):
        return int(self.lib.dt_track_row_width(self.h))

    def track_detect(self, frames):
        """frames [F,H,W,3] -> z rows [F,G,G,row_width] = [conv_feat | x_bbox | pad] (frame-shard half 1)."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dim() == 4
        F = frames.shape[0]
        gh, gw = self.grid
        z = self._f32(F, gh, gw, self.lib.dt_track_row_width(self.h))
        self._sync_stream()
        self._check(self.lib.dt_track_detect(self.h, _dptr(frames), self._frames_dtype(frames), F, _dptr(z)), "dt_track_detect")
        return z

    def track_recurrent(self, z, want_det=False):
        """z [n_clips,T,G,G,row_width] -> tracking grid [n_clips,T,G,G,NB,5+C] (frame-shard half 2)."""
        assert z.is_cuda and z.is_contiguous() and z.dim() == 5 and z.dtype == self.torch.float32
        n_clips, T = z.shape[:2]
        gh, gw = self.grid
        trk = self._f32(n_clips, T, gh, gw, self.nb_box, 5 + self.nb_class)
        det = self._f32(n_clips, T, gh, gw, self.nb_box, 5 + self.nb_class) if want_det else None
        self._sync_stream()
        self._check(self.lib.dt_track_recurrent(self.h, _dptr(z), n_clips, T, _dptr(trk), _dptr(det)), "dt_track_recurrent")
        return (trk, det) if want_det else trk

    def track_xproj_width(self):
        return int(self.lib.dt_track_xproj_width(self.h))

    def track_detect_xproj(self, frames):
        """frames [F,H,W,3] -> ConvLSTM input-projection rows [F,G,G,4U] (frame-shard half 1, projection included)."""
        assert frames.is_cuda and frames.is_contiguous() and frames.dim() == 4
        F = frames.shape[0]
        gh, gw = self.grid
        xp = self._f32(F, gh, gw, self.track_xproj_width())
        self._sync_stream()
        self._check(self.lib.dt_track_detect_xproj(self.h, _dptr(frames), self._frames_dtype(frames), F, _dptr(xp), None),
                    "dt_track_detect_xproj")
        return xp

    def track_recurrent_xproj(self, xp):
        """xp [n_clips,T,G,G,4U] -> tracking grid [n_clips,T,G,G,NB,5+C] (frame-shard half 2: the recurrence alone)."""
        assert xp.is_cuda and xp.is_contiguous() and xp.dim() == 5 and xp.dtype == self.torch.float32
        n_clips, T = xp.shape[:2]
        gh, gw = self.grid
        trk = self._f32(n_clips, T, gh, gw, self.nb_box, 5 + self.nb_class)
        self._sync_stream()
        self._check(self.lib.dt_track_recurrent_xproj(self.h, _dptr(xp), n_clips, T, _dptr(trk)), "dt_track_recurrent_xproj")
        return trk

    # ---- tiny tracker ---------------------------------------------------
    def tiny_load(self, D, units, kernel, recurrent, bias, dense_kernel, dense_bias):
        ks = [_hptr(a) for a in (kernel, recurrent, bias, dense_kernel, dense_bias)]
        self.tiny_out = int(np.asarray(dense_kernel).shape[1])
        self.tiny_D = int(D)
        self._check(self.lib.dt_tiny_load(self.h, D, units, self.tiny_out, *[k[1] for k in ks]), "dt_tiny_load")

    def heatmap_from_boxes(self, box4, hmap_size):
        """box4 [n,4] centre-format -> [n, hs*hs] 0/1 heatmaps (utils.generate_heatmap_feat)."""
        n = box4.shape[0]
        out = self._f32(n, hmap_size * hmap_size)
        self._sync_stream()
        self._check(self.lib.dt_heatmap_from_boxes(self.h, _dptr(box4), n, hmap_size, _dptr(out)), "dt_heatmap_from_boxes")
        return out

    def heatmap_from_xywh64(self, xywh, hmap_size):
        """xywh [n,4] float64 (det_x, det_y, det_w, det_h) exactly as utils.generate_heatmap_feat takes them."""
        assert xywh.dtype == self.torch.float64 and xywh.is_cuda and xywh.is_contiguous()
        n = xywh.shape[0]
        out = self._f32(n, hmap_size * hmap_size)
        self._sync_stream()
        self._check(self.lib.dt_heatmap_from_xywh64(self.h, _dptr(xywh), n, hmap_size, _dptr(out)), "dt_heatmap_from_xywh64")
        return out

    def encode_targets(self, objs, counts, dims, aug, grid_h, grid_w, nb_box, nb_class, image_h, image_w,
                       true_box_buffer, anchors):
        """objs int32 [n,cap,5], counts int32 [n], dims int32 [n,2], aug float64 [n,4] or None ->
        (y float64 [n,GH,GW,NB,5+C], b float64 [n,TBB,4])  (preprocessing.py:171-188, 214-293)."""
        t = self.torch
        assert objs.dtype == t.int32 and counts.dtype == t.int32 and dims.dtype == t.int32
        assert objs.is_cuda and objs.is_contiguous() and counts.is_contiguous() and dims.is_contiguous()
        n, cap = objs.shape[0], objs.shape[1]
        if aug is not None:
            assert aug.dtype == t.float64 and aug.is_cuda and aug.is_contiguous() and aug.shape == (n, 4)
        an = np.ascontiguousarray(anchors, dtype=np.float64)
        assert an.size == 2 * nb_box
        y = t.empty((n, grid_h, grid_w, nb_box, 5 + nb_class), dtype=t.float64, device=self.device)
        b = t.empty((n, true_box_buffer, 4), dtype=t.float64, device=self.device)
        self._sync_stream()
        self._check(self.lib.dt_encode_targets(self.h, _dptr(objs), _dptr(counts), _dptr(dims),
                                               _dptr(aug) if aug is not None else None, n, cap, grid_h, grid_w,
                                               nb_box, nb_class, image_h, image_w, true_box_buffer,
                                               an.ctypes.data_as(ctypes.c_void_p), _dptr(y), _dptr(b)),
                    "dt_encode_targets")
        return y, b

    def rect_from_heatmap(self, heat, hmap_size, thresh=0.75):
        """heat [n, hs*hs] -> int32 [n,4] (x1,y1,x2,y2) (utils.generate_rectangle_from_heatmap)."""
        t = self.torch
        n = heat.shape[0]
        out = t.empty((n, 4), dtype=t.int32, device=self.device)
        self._sync_stream()
        self._check(self.lib.dt_rect_from_heatmap(self.h, _dptr(heat), n, hmap_size, float(thresh), _dptr(out)),
                    "dt_rect_from_heatmap")
        return out

    def tiny_forward(self, feat, det, pool="Global"):
        """feat [n_seq,T,fh,fw,fc], det [n_seq,T,4] -> [n_seq,T,4]."""
        t = self.torch
        assert feat.is_cuda and feat.is_contiguous() and feat.dtype == t.float32
        assert det.is_cuda and det.is_contiguous() and det.dtype == t.float32
        n_seq, T, fh, fw, fc = feat.shape
        out = self._f32(n_seq, T, self.tiny_out)
        self._sync_stream()
        self._check(self.lib.dt_tiny_forward(self.h, _dptr(feat), _dptr(det), n_seq, T, fh, fw, fc,
                                             0 if pool == "Global" else 1, _dptr(out)), "dt_tiny_forward")
        return out

    def tiny_features(self, feat, det, D, pool="Global"):
        """feat [n,fh,fw,fc], det [n,4] -> rows [n,D] (pooled feature (+) det box)."""
        n, fh, fw, fc = feat.shape
        x = self._f32(n, D)
        self._sync_stream()
        self._check(self.lib.dt_tiny_features(self.h, _dptr(feat), _dptr(det), n, fh, fw, fc,
                                              0 if pool == "Global" else 1, _dptr(x)), "dt_tiny_features")
        return x

    def tiny_sequence(self, x):
        """x [n_seq,T,D] -> [n_seq,T,4]."""
        assert x.is_cuda and x.is_contiguous()
        n_seq, T, _ = x.shape
        out = self._f32(n_seq, T, self.tiny_out)
        self._sync_stream()
        self._check(self.lib.dt_tiny_sequence(self.h, _dptr(x), n_seq, T, _dptr(out)), "dt_tiny_sequence")
        return out

    def top_box(self, boxes, counts):
        """boxes [F,cap,8], counts [F] -> [F,4] highest-score box per frame (zeros if none)."""
        F, cap, _ = boxes.shape
        out = self._f32(F, 4)
        self._sync_stream()
        self._check(self.lib.dt_top_box(self.h, _dptr(boxes), _dptr(counts), F, cap, _dptr(out)), "dt_top_box")
        return out

    # ---- layer-level (parity tests) -----------------------------------
    def conv2d(self, x, kernel_hwio, bias=None, leaky_slope=1.0, pool=0):
        t = self.torch
        assert x.is_cuda and x.is_contiguous() and x.dtype == t.float32
        B, H, W, Cin = x.shape
        k, _, ci, Cout = kernel_hwio.shape
        assert ci == Cin
        kk, kp = _hptr(kernel_hwio)
        bb, bp = _hptr(bias) if bias is not None else (None, None)
        if pool == 0:
            out, out2 = self._f32(B, H, W, Cout), None
        elif pool == 1:
            out, out2 = self._f32(B, H // 2, W // 2, Cout), None
        elif pool == 2:
            out, out2 = self._f32(B, H, W, Cout), self._f32(B, H // 2, W // 2, Cout)
        else:
            out, out2 = self._f32(B, H // 2, W // 2, 4 * Cout), None
        self._sync_stream()
        self._check(self.lib.dt_conv2d(self.h, _dptr(x), B, H, W, Cin, kp, k, Cout, bp, float(leaky_slope), pool,
                                       _dptr(out), _dptr(out2)), "dt_conv2d")
        return (out, out2) if pool == 2 else out

    def convlstm_step(self, x, h, c, kernel, recurrent, bias):
        t = self.torch
        B, H, W, Cx = x.shape
        U = h.shape[-1]
        ho, co = t.empty_like(h), t.empty_like(c)
        ks = [_hptr(a) for a in (kernel, recurrent, bias)]
        self._sync_stream()
        self._check(self.lib.dt_convlstm_step(self.h, _dptr(x), B, H, W, Cx, _dptr(h), _dptr(c), U,
                                              ks[0][1], ks[1][1], ks[2][1], _dptr(ho), _dptr(co)), "dt_convlstm_step")
        return ho, co

    def gemm_split_bf16(self, v, u, half=0, nt=3):
        """v [P,Mt,K], u [P,N,K] float32 -> [P,Mt,N] through wino_gemm_s3.hip: the Winograd-domain contraction at a caller-chosen
        shape.  nt=3: three-term bf16 split of both operands on the device, six MFMA partial products per multiply; nt=2: the
        fp16 form (two terms of the scaled operands, three products)."""
        t = self.torch
        assert v.is_cuda and u.is_cuda and v.is_contiguous() and u.is_contiguous() and v.dtype == t.float32 and u.dtype == t.float32
        P, Mt, K = v.shape
        assert u.shape[0] == P and u.shape[2] == K
        N = u.shape[1]
        out = self._f32(P, Mt, N)
        self._sync_stream()
        self._check(self.lib.dt_gemm_split(self.h, _dptr(v), _dptr(u), P, Mt, K, N, int(half), int(nt), _dptr(out)), "dt_gemm_split")
        return out

    def gemm_split(self, v, u, half=0, nt=2):
        return self.gemm_split_bf16(v, u, half=half, nt=nt)

    # ---- profiling ------------------------------------------------------
    def profile_enable(self, on=True):
        self._sync_stream()
        self._check(self.lib.dt_profile_enable(self.h, 1 if on else 0), "dt_profile_enable")

    def profile_reset(self):
        self._check(self.lib.dt_profile_reset(self.h), "dt_profile_reset")

    def profile_names(self):
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self.lib.dt_profile_names(self.h, buf, len(buf)), "dt_profile_names")
        return [n for n in buf.value.decode().split("\n") if n]

    def reload_policy(self):
        """re-read the DT_* tuning / test knobs from the environment (they are read once, in dt_create)"""
        self._check(self.lib.dt_policy_reload(self.h), "dt_policy_reload")

    def policy_set(self, name, value):
        """one knob of THIS context ("pin"), without the process environment (dt_policy_set)"""
        self._check(self.lib.dt_policy_set(self.h, name.encode(), int(value)), "dt_policy_set")

    def graph_enable(self, on=True):
        """hipGraph replay of the detector trunk and the ConvLSTM recurrence (low-latency serving)."""
        self._check(self.lib.dt_graph_enable(self.h, 1 if on else 0), "dt_graph_enable")

    def profile_read(self, name):
        n = ctypes.c_int64(0)
        ms, fl, by = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self.lib.dt_profile_read(self.h, name.encode(), ctypes.byref(n), ctypes.byref(ms),
                                             ctypes.byref(fl), ctypes.byref(by)), "dt_profile_read")
        return dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)


_default_ctx = None


def default_context():
    """Process-wide context on the current device (created on first use)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
