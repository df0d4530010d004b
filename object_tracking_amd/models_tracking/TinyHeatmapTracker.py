"""TinyHeatmapTracker -- the reference's heatmap variant of the single-object
LSTM tracker (models_tracking/TinyHeatmapTracker.py:16-48) on the MI355X.

Same graph as TinyTracker with the detection input and the output replaced by
flattened HEATMAP_SIZE x HEATMAP_SIZE occupancy maps: pooled detector feature
(+) heatmap(det box) -> LSTM(512, implementation=2) -> TimeDistributed(Dense(
HEATMAP_SIZE**2, sigmoid)).  Heatmaps are built exactly like the data generator
does (utility/utils.py:53-58 called as in preprocessing.py:455) and boxes are
read back with generate_rectangle_from_heatmap (utility/utils.py:61-79); both
run on the device (dt_heatmap_from_boxes / dt_rect_from_heatmap).  The LSTM
step is the TinyTracker kernel; the 1024-wide Dense head is a 1x1 MFMA GEMM with
a sigmoid epilogue.
"""
from models_tracking.TinyTracker import NativeTinyModel, TinyTracker


class TinyHeatmapTracker(TinyTracker):
    def __init__(self, config=None, feature_dims=(26, 26, 512), weights=None, ctx=None):
        cfg = config
        if cfg is None:
            from models_tracking.BaseTracker import DEFAULT_CONFIG
            cfg = DEFAULT_CONFIG
        self.HEATMAP_SIZE = cfg["model_tracker"].get("heatmap_size", 32)
        super(TinyHeatmapTracker, self).__init__(config, feature_dims, weights, ctx)

    def det_width(self):
        return self.HEATMAP_SIZE ** 2

    def load_tracker_model(self):
        self.model_tracker = NativeTinyModel(self, self._ctx)
        if self._weights is not None:
            self.model_tracker.set_weights(self._weights)
        print("NativeTinyModel (heatmap %dx%d):" % (self.HEATMAP_SIZE, self.HEATMAP_SIZE), end=" ")
        self.model_tracker.summary()

    def frame_rows(self, frames, detector):
        """frames [F,H,W,3] -> (rows [F, feat + hs*hs], det4 [F,4]); the detection box is
        rasterised into the heatmap input on the device."""
        ctx = self.model_tracker.ctx
        assert detector.model.ctx is ctx, "construct TinyHeatmapTracker(ctx=detector.model.ctx)"
        d = detector.model.to_device(frames)
        F = d.shape[0]
        ctx.detect_forward_internal(d)
        feat = ctx.detector_tap("act_13", F)
        gh, gw = ctx.grid
        netout = ctx.detector_tap("conv_23", F).reshape(F, gh, gw, ctx.nb_box, 5 + ctx.nb_class)
        r = ctx.decode(netout, detector.OBJ_THRESHOLD, detector.NMS_THRESHOLD, detector.ANCHORS,
                       len(detector.LABELS))      # full cap: boxes come in (row, col, anchor) order, the
        # best-scoring one may sit behind MAX_BOX_PER_IMAGE others
        det4 = ctx.top_box(r["boxes"], r["counts"])
        heat = ctx.heatmap_from_boxes(det4, self.HEATMAP_SIZE)
        return ctx.tiny_features(feat, heat, self.feature_width() + self.det_width(), self.pool), det4

    def track_sequences(self, frames, detector, thresh=0.75):
        """frames [n_seq,T,H,W,3] -> (heatmaps [n_seq,T,hs*hs], rects int32 [n_seq,T,4] = x1,y1,x2,y2
        in heatmap cells, (hs,hs,-1,-1) where nothing reaches `thresh`)."""
        n_seq, T = frames.shape[:2]
        flat = frames.reshape((n_seq * T,) + tuple(frames.shape[2:]))
        rows, _ = self.frame_rows(flat, detector)
        ctx = self.model_tracker.ctx
        heat = ctx.tiny_sequence(rows.reshape(n_seq, T, -1).contiguous())
        rects = ctx.rect_from_heatmap(heat.reshape(n_seq * T, -1), self.HEATMAP_SIZE, thresh)
        return heat, rects.reshape(n_seq, T, 4)
