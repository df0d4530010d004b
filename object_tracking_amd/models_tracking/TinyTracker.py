"""TinyTracker -- the reference's ROLO-style single-object LSTM tracker
(models_tracking/TinyTracker.py:16-41) on the MI355X.

Graph (TinyTracker.py:25-41): per time step GlobalMaxPooling2D over the detector
feature map (pool == 'Global'; 'Max' = MaxPooling2D(4,4)+Flatten) concatenated
with the 4-vector detection box -> LSTM(512, implementation=2,
return_sequences=True) -> TimeDistributed(Dense(4, sigmoid)).  The reference has
no inference entry point for it (only fit_generator); `model_tracker.predict`
below is the forward its layer list defines, batched over any number of
independent tracks.
"""
import numpy as np

import mi355_dt
from models_tracking.BaseTracker import BaseTracker


class NativeTinyModel(object):
    """Replaces the Keras `Model([img_input, det_input], output)` (TinyTracker.py:39)."""

    def __init__(self, owner, ctx=None):
        self.owner = owner
        self.ctx = ctx if ctx is not None else mi355_dt.Context()
        self.loaded = False

    def set_weights(self, w):
        """dict(kernel [D,4U], recurrent [U,4U], bias [4U], dense_kernel [U,4],
        dense_bias [4]) in Keras layouts ('recurrent_layer', 'output')."""
        D, n4 = w["kernel"].shape
        self.ctx.tiny_load(int(D), int(n4 // 4), w["kernel"], w["recurrent"], w["bias"], w["dense_kernel"],
                           w["dense_bias"])
        self.loaded = True

    def forward(self, feat, det):
        import torch
        dev = self.ctx.device
        if isinstance(feat, np.ndarray):
            feat = torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32))
        if isinstance(det, np.ndarray):
            det = torch.from_numpy(np.ascontiguousarray(det, dtype=np.float32))
        return self.ctx.tiny_forward(feat.to(dev).contiguous(), det.to(dev).contiguous(), pool=self.owner.pool)

    def predict(self, inputs, batch_size=None):
        """[img_fv (B,T,w,h,c), det (B,T,4)] -> (B,T,4) numpy."""
        return self.forward(inputs[0], inputs[1]).cpu().numpy()

    def summary(self):
        o = self.owner
        print("NativeTinyModel: pool=%s(%dx%dx%d) (+) det4 -> LSTM(%d) -> Dense(4,sigmoid); T=%d on %s" % (
            o.pool, o._w, o._h, o._c, o.LSTM_UNITS, o.SEQUENCE_LENGTH, self.ctx.device))


class TinyTracker(BaseTracker):
    def __init__(self, config=None, feature_dims=(26, 26, 512), weights=None, ctx=None):
        super(TinyTracker, self).__init__(config, feature_dims)
        self.LSTM_UNITS = self.config["model_tracker"]["lstm_units"]
        self.SEQUENCE_LENGTH = self.config["model_tracker"]["sequence_length"]
        self._weights = weights
        self._ctx = ctx
        self.load_tracker_model()

    def det_width(self):
        return 4

    def feature_width(self):
        if self.pool == 'Global':
            return self._c
        return (self._w // 4) * (self._h // 4) * self._c

    def load_tracker_model(self):
        self.model_tracker = NativeTinyModel(self, self._ctx)
        if self._weights is not None:
            self.model_tracker.set_weights(self._weights)
        self.model_tracker.summary()

    # ------------------------------------------------------------------
    # Inference pipeline (addition: the reference only trains this model).  The
    # detector is this build's KerasYOLO sharing the tracker's dt_ctx; the feature
    # layer is its 'act_13' tap (26x26x512 at 416x416, config.json:9 fv_layer) and
    # the detection box is the highest-score survivor of the frame as (cx,cy,w,h)
    # in image-relative units (the training-time generator takes the first
    # detection matching the ground-truth label, preprocessing.py:421-456).
    def frame_rows(self, frames, detector):
        """frames [F,H,W,3] (numpy / device) -> (rows [F,D] = pooled feature (+) det box, det4 [F,4])."""
        ctx = self.model_tracker.ctx
        assert detector.model.ctx is ctx, "construct TinyTracker(ctx=detector.model.ctx)"
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
        return ctx.tiny_features(feat, det4, self.feature_width() + self.det_width(), self.pool), det4

    def track_sequences(self, frames, detector):
        """frames [n_seq,T,H,W,3] -> tracked boxes [n_seq,T,4] (device tensor)."""
        n_seq, T = frames.shape[:2]
        flat = frames.reshape((n_seq * T,) + tuple(frames.shape[2:]))
        rows, _ = self.frame_rows(flat, detector)
        return self.model_tracker.ctx.tiny_sequence(rows.reshape(n_seq, T, -1).contiguous())
