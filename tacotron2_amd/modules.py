"""Mirrors of the primitive layers of the reference's tacotron/modules.py that lie on the V2
mel-generation path: Embedding (:46-70), PreNet (:73-98), Conv1d (:133-160), ZoneoutLSTMCell
(:235-283).  The objects hold sizes and (optionally) weights in TensorFlow layouts; they hold no
arithmetic: the assemblies in tacotron_v2.py hand everything to the HIP library (engine.Engine).
HighwayNet / CBHG (:101-232) belong to the Tacotron-1 path: tacotron_v1.py (t2v1_cbhg)."""
from __future__ import annotations

import numpy as np

from .rnn_impl import LSTMImpl, lstm_cell_factory


class Embedding:
    """modules.py:46-70.  Calling it outside an assembly does the table lookup on the host after
    the reference's range assertion (:65-66); inside TacotronV2Model the lookup runs on the GPU
    (t2_encode)."""

    def __init__(self, num_symbols, embedding_dim, index_offset=0, weights=None, name=None):
        self._num_symbols = num_symbols
        self._embedding_dim = embedding_dim
        self.index_offset = int(index_offset)
        self.weights = weights          # {"embedding": [num_symbols, embedding_dim]}

    def __call__(self, inputs):
        ids = np.asarray(inputs)
        if np.any(ids < self.index_offset) or np.any(ids >= self.index_offset + self._num_symbols):
            raise AssertionError("embedding id out of range")      # tf.assert_greater_equal / tf.assert_less
        if self.weights is None:
            raise ValueError("Embedding called without weights")
        return np.asarray(self.weights["embedding"], dtype=np.float32)[ids - self.index_offset]

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self._embedding_dim)


class PreNet:
    """modules.py:73-98: Dense(out_units, relu) + dropout(drop_rate) when is_training or apply_dropout_on_inference.
    Bound to a decoder, active dropout is sampled inside the training step; called on its own (``weights`` given), the dropout
    mask comes from the library's counter-based generator: u01(seed, stream, time, row, column) with ``seed`` / ``stream`` /
    ``time`` attributes of this object (``time`` advances per call; TensorFlow's random stream cannot be reproduced)."""

    def __init__(self, out_units, is_training, drop_rate=0.5, apply_dropout_on_inference=False, weights=None, name=None):
        self.out_units = out_units
        self.drop_rate = drop_rate
        self.is_training = is_training
        self.apply_dropout_on_inference = apply_dropout_on_inference
        self.weights = weights          # {"kernel": [in, out], "bias": [out]}
        self.seed, self.stream, self.time = 0, 1, 0

    @property
    def dropout_enabled(self):
        return self.is_training or self.apply_dropout_on_inference

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.out_units,)

    def __call__(self, inputs, **kwargs):
        return self.call(inputs)

    def call(self, inputs, **kwargs):
        """Dense(out_units, relu) of modules.py:88-91 on the HIP library (t2_dense), then tf.layers.dropout(training=
        dropout_enabled) (modules.py:90,96-98; t2_dropout).  Stand-alone use needs ``weights``."""
        import torch
        from . import _lib
        from .engine import shared_context
        if self.weights is None:
            raise ValueError("PreNet called without weights")
        x = inputs if isinstance(inputs, torch.Tensor) and inputs.is_cuda else None
        ctx = shared_context(x.device.index if x is not None else torch.cuda.current_device())
        x = ctx.f32(inputs)
        k, b = ctx.f32(self.weights["kernel"]), ctx.f32(self.weights["bias"])
        rows = int(np.prod(x.shape[:-1]))
        y = torch.empty(tuple(x.shape[:-1]) + (self.out_units,), dtype=torch.float32, device=ctx.device)
        with ctx.ordered():
            _lib.check(ctx.lib.t2_dense(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), rows, int(x.shape[-1]), self.out_units, 1, y.data_ptr()))
            if self.dropout_enabled and self.drop_rate > 0:
                _lib.check(ctx.lib.t2_dropout(ctx.handle, y.data_ptr(), rows, self.out_units, float(self.drop_rate), int(self.seed) & 0xFFFFFFFF,
                                              int(self.stream), int(self.time), y.data_ptr()))
                self.time += 1
        return y


class Conv1d:
    """modules.py:133-160: conv1d(SAME, no bias) -> batch norm -> activation -> dropout."""

    def __init__(self, kernel_size, out_channels, activation, is_training, use_bias=False, drop_rate=0.0,
                 weights=None, name=None):
        if use_bias:
            raise NotImplementedError("the path never uses a conv bias (modules.py:136 default False)")
        self.kernel_size = kernel_size
        self.out_channels = out_channels
        self.activation = activation
        self.is_training = is_training
        self.drop_rate = drop_rate
        self.weights = weights
        self.name = name


class ZoneoutLSTMCell:
    """modules.py:235-283.  Same constructor checks: zoneout factors outside [0, 1] raise
    ValueError (:241-245); a non-tuple state raises NotImplementedError (:268-269)."""

    def __init__(self, num_units, is_training, zoneout_factor_cell=0.0, zoneout_factor_output=0.0, state_is_tuple=True,
                 lstm_impl=LSTMImpl.LSTMCell, weights=None, name=None):
        zm = min(zoneout_factor_output, zoneout_factor_cell)
        zs = max(zoneout_factor_output, zoneout_factor_cell)
        if zm < 0. or zs > 1.:
            raise ValueError('One/both provided Zoneout factors are not in [0, 1]')
        self._cell = lstm_cell_factory(lstm_impl, num_units)
        self._zoneout_cell = zoneout_factor_cell
        self._zoneout_outputs = zoneout_factor_output
        self.is_training = is_training
        self.state_is_tuple = state_is_tuple
        self.weights = weights          # {"kernel": [in + h, 4h], "bias": [4h]}
        self.seed, self.stream, self.time = 0, 10, 0     # counter-based generator coordinates of the sampled branch (time advances per call)

    @property
    def num_units(self):
        return self._cell.num_units

    @property
    def state_size(self):
        return self._cell.state_size

    @property
    def output_size(self):
        return self._cell.output_size

    def zero_state(self, batch_size, dtype=None, device="cuda"):
        import torch
        z = lambda: torch.zeros((batch_size, self.num_units), dtype=torch.float32, device=device)
        return (z(), z())                       # LSTMStateTuple(c, h)

    def __call__(self, inputs, state, scope=None):
        """modules.py:261-283 on the HIP library: returns (raw output h, (zoned c, zoned h)).  Stand-alone use needs ``weights``.
        ``is_training`` with non-zero factors runs the sampled branch (:274-276; t2_zoneout_lstm_cell_train) with masks
        u01(seed, stream | stream + 1, time, row, unit) of this object's ``seed`` / ``stream`` / ``time``."""
        import torch
        from . import _lib
        from .engine import shared_context
        if not self.state_is_tuple:
            raise NotImplementedError("non-tuple state is not implemented")
        sampled = self.is_training and (self._zoneout_cell > 0 or self._zoneout_outputs > 0)
        if self.weights is None:
            raise ValueError("ZoneoutLSTMCell called without weights")
        dev = inputs.device.index if isinstance(inputs, torch.Tensor) and inputs.is_cuda else torch.cuda.current_device()
        ctx = shared_context(dev)
        x, c, h = ctx.f32(inputs), ctx.f32(state[0]), ctx.f32(state[1])
        kernel, bias = ctx.f32(self.weights["kernel"]), ctx.f32(self.weights["bias"])
        B, H = int(x.shape[0]), self.num_units
        xh = torch.cat([x, h], dim=-1).contiguous()
        if tuple(kernel.shape) != (xh.shape[1], 4 * H):
            raise ValueError(f"kernel shape {tuple(kernel.shape)} does not fit inputs + state ({xh.shape[1]}, {4 * H})")
        z = torch.empty((B, 4 * H), dtype=torch.float32, device=ctx.device)
        out, c2, h2 = (torch.empty((B, H), dtype=torch.float32, device=ctx.device) for _ in range(3))
        with ctx.ordered():
            if sampled:
                _lib.check(ctx.lib.t2_zoneout_lstm_cell_train(ctx.handle, xh.data_ptr(), c.data_ptr(), h.data_ptr(), kernel.data_ptr(), bias.data_ptr(),
                                                              B, int(xh.shape[1]), H, float(self._zoneout_cell), float(self._zoneout_outputs), 1.0,
                                                              int(self.seed) & 0xFFFFFFFF, int(self.stream), int(self.time),
                                                              z.data_ptr(), out.data_ptr(), c2.data_ptr(), h2.data_ptr()))
                self.time += 1
            else:
                _lib.check(ctx.lib.t2_zoneout_lstm_cell(ctx.handle, xh.data_ptr(), c.data_ptr(), h.data_ptr(), kernel.data_ptr(), bias.data_ptr(),
                                                        B, int(xh.shape[1]), H, float(self._zoneout_cell), float(self._zoneout_outputs), 1.0,
                                                        z.data_ptr(), out.data_ptr(), c2.data_ptr(), h2.data_ptr()))
        return out, (c2, h2)
