"""Mirrors of the primitive layers of the reference's tacotron/modules.py that lie on the V2
mel-generation path: Embedding (:46-70), PreNet (:73-98), Conv1d (:133-160), ZoneoutLSTMCell
(:235-283).  The objects hold sizes and (optionally) weights in TensorFlow layouts; they hold no
arithmetic: the assemblies in tacotron_v2.py hand everything to the HIP library (engine.Engine).
HighwayNet / CBHG (:101-232) belong to the Tacotron-1 path and are not built yet."""
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
    """modules.py:73-98: Dense(out_units, relu) + dropout(drop_rate) when is_training or
    apply_dropout_on_inference.  The HIP path implements the dropout-free arithmetic; a request for
    active dropout is rejected when the prenet is bound to a decoder."""

    def __init__(self, out_units, is_training, drop_rate=0.5, apply_dropout_on_inference=False, weights=None, name=None):
        self.out_units = out_units
        self.drop_rate = drop_rate
        self.is_training = is_training
        self.apply_dropout_on_inference = apply_dropout_on_inference
        self.weights = weights          # {"kernel": [in, out], "bias": [out]}

    @property
    def dropout_enabled(self):
        return self.is_training or self.apply_dropout_on_inference

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[:-1]) + (self.out_units,)


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

    @property
    def num_units(self):
        return self._cell.num_units

    @property
    def state_size(self):
        return self._cell.state_size

    @property
    def output_size(self):
        return self._cell.output_size

    def __call__(self, inputs, state, scope=None):
        if not self.state_is_tuple:
            raise NotImplementedError("non-tuple state is not implemented")
        raise RuntimeError("ZoneoutLSTMCell runs inside a decoder / encoder assembly (see tacotron_v2.py); "
                           "a stand-alone cell call is not exposed by the HIP library")
