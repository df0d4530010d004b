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

    def feature_width(self):
        if self.pool == 'Global':
            return self._c
        return (self._w // 4) * (self._h // 4) * self._c

    def load_tracker_model(self):
        self.model_tracker = NativeTinyModel(self, self._ctx)
        if self._weights is not None:
            self.model_tracker.set_weights(self._weights)
        self.model_tracker.summary()
