"""Mirror of the reference's tacotron/rnn_wrappers.py: the RNNCell decorators that compose one
decoder step (prenet -> attention-wrapped RNN -> concat context -> mel / stop projections).

Every class keeps the reference's protocol (``state_size``, ``output_size``, ``zero_state``,
``compute_output_shape``, ``call`` / ``__call__``).  The HIP library runs the cell stack as ONE unit
(``t2_decode_step``, include/t2mi.h), so the composition is a description that the outermost wrapper in
use binds to the library: ``OutputAndStopTokenWrapper.call`` is one step; ``DecoderRNNV2.call``,
``AttentionRNN.call`` and the two inner wrappers of an ``AttentionRNN`` run the same step (bound through
outer layers nobody reads) and return what the reference's cell returns at that level -- the raw output
of the last decoder LSTM, ``concat(attention cell output, new context)``, the attention cell output.
State is a ``DecoderState``: ``state.get(name)`` reads any field of the reference's state tuple.
"""
from __future__ import annotations

from math import gcd
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .hparams import T2Dims
from .modules import PreNet, ZoneoutLSTMCell
from .weights import init_weights


class DeferredState:
    """zero_state of a cell that is used on its own, before the width of its input is known (the reference's zero_state takes
    only the batch size): becomes a real DecoderState at the first ``call``."""

    def __init__(self, batch_size):
        self.batch_size, self._real = batch_size, None

    def bind(self, make):
        if self._real is None:
            self._real = make()
        return self._real

    def get(self, name):
        if self._real is None:
            raise RuntimeError("the cell has not been called yet: every field of its zero state is zero")
        return self._real.get(name)


def real_state(state):
    return state._real if isinstance(state, DeferredState) else state


class _Delegating:
    """state_size / output_size / zero_state delegation shared by the wrappers."""

    @property
    def state_size(self):
        return self._cell.state_size

    @property
    def output_size(self):
        return self._cell.output_size

    def zero_state(self, batch_size, dtype=None):
        return self._cell.zero_state(batch_size, dtype)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.output_size)

    def __call__(self, inputs, state):
        return self.call(inputs, state)


class AttentionWrapperStateSize:
    """The fields of tf.contrib.seq2seq.AttentionWrapperState that the reference reads
    (rnn_wrappers.py:68 uses ``state_size.attention``)."""

    def __init__(self, cell_state, attention, alignments):
        self.cell_state, self.attention, self.time = cell_state, attention, ()
        self.alignments, self.alignment_history, self.attention_state = alignments, (), alignments


class _AttentionWrapperSpec(_Delegating):
    """Stands for tf.contrib.seq2seq.AttentionWrapper(cell, mechanism, alignment_history=True,
    output_attention=False) as built at rnn_wrappers.py:118-122."""

    def __init__(self, cell, attention_mechanism):
        self._cell_spec = cell
        self.mechanism = attention_mechanism

    @property
    def state_size(self):
        m = self.mechanism
        return AttentionWrapperStateSize(self._cell_spec.state_size, m.values_depth, m.alignments_size)

    @property
    def output_size(self):
        return self._cell_spec.output_size      # output_attention=False

    def zero_state(self, batch_size, dtype=None):
        return _owner(self).zero_state(batch_size, dtype)

    def call(self, inputs, state):
        raise RuntimeError("tf.contrib.seq2seq.AttentionWrapper takes concat(prenet output, previous context); in this package the "
                           "PreNets and the attention cell run as one unit: call the owning DecoderPreNetWrapper / AttentionRNN")


def _owner(wrapper):
    owner = getattr(wrapper, "_owner", None)
    if owner is None:
        raise RuntimeError("this wrapper is bound through the AttentionRNN that built it (rnn_wrappers.py:118-127); construct an "
                           "AttentionRNN(cell, prenets, attention_mechanism) and use its inner wrappers")
    return owner


class DecoderPreNetWrapper(_Delegating):
    """rnn_wrappers.py:30-53: ``call`` = PreNets, then the attention-wrapped cell; returns the attention cell's raw output."""

    def __init__(self, cell, prenets: Tuple[PreNet, ...]):
        self._cell = cell
        self.prenets = tuple(prenets)

    def zero_state(self, batch_size, dtype=None):
        return _owner(self).zero_state(batch_size, dtype)

    def call(self, inputs, state):
        _owner(self)._step(inputs, state)
        return state.get("attention_output"), state


class ConcatOutputAndAttentionWrapper(_Delegating):
    """rnn_wrappers.py:56-78: output_size = cell output + state_size.attention (:66-68)."""

    def __init__(self, cell):
        self._cell = cell

    @property
    def output_size(self):
        return self._cell.output_size + self._cell.state_size.attention

    def zero_state(self, batch_size, dtype=None):
        return _owner(self).zero_state(batch_size, dtype)

    def call(self, inputs, state):
        return _owner(self).call(inputs, state)


class AttentionRNN(_Delegating):
    """rnn_wrappers.py:112-144: ``cell`` is any LSTM cell description (ZoneoutLSTMCell here),
    ``prenets`` a tuple of PreNet, ``attention_mechanism`` a LocationSensitiveAttention."""

    def __init__(self, cell: ZoneoutLSTMCell, prenets: Tuple[PreNet, ...], attention_mechanism, name=None, dtype=None):
        attention_cell = _AttentionWrapperSpec(cell, attention_mechanism)
        prenet_cell = DecoderPreNetWrapper(attention_cell, prenets)     # prenet -> attention
        concat_cell = ConcatOutputAndAttentionWrapper(prenet_cell)      # prenet -> attention -> concat
        self._cell = concat_cell
        self.lstm, self.prenets, self.mechanism = cell, tuple(prenets), attention_mechanism
        attention_cell._owner = prenet_cell._owner = concat_cell._owner = self
        self.input_units = None         # width of the decoder input (num_mels * n_feed_frame), known at the first call
        self.weights = None             # optional {"decoder/...", "attention/...": array} for a cell used on its own
        self._decoder = None

    # -- RNNCell protocol on the cell itself (rnn_wrappers.py:143-144) -----------------------------------------------------
    def _bound_decoder(self):
        """A bare AttentionRNN is bound through a DecoderRNNV2 with two small LSTM layers nobody reads: the attention cell,
        its PreNets and the attention mechanism do not depend on them."""
        if self._decoder is None:
            from .tacotron_v2 import DecoderRNNV2
            self._decoder = DecoderRNNV2(4, self, is_training=self.lstm.is_training, zoneout_factor_cell=self.lstm._zoneout_cell,
                                         zoneout_factor_output=self.lstm._zoneout_outputs)
            own = ("decoder/prenet_", "decoder/attention_lstm/", "attention/")
            self._decoder.weights = {k: v for k, v in (self.weights or {}).items() if k.startswith(own)}
        return self._decoder

    def zero_state(self, batch_size, dtype=None):
        return self._bound_decoder().zero_state(batch_size, dtype)

    def _step(self, inputs, state):
        self.input_units = int(inputs.shape[-1])
        self._bound_decoder().call(inputs, state)

    def call(self, inputs, state):
        """prenet -> attention-wrapped cell -> concat(cell output, NEW context) (rnn_wrappers.py:76-78): [B, A + E]."""
        self._step(inputs, state)
        return torch.cat([state.get("attention_output"), state.get("attention")], dim=-1), state


class OutputProjectionWrapper(_Delegating):
    """rnn_wrappers.py:147-186 (Tacotron-1 only): ``cell`` followed by a linear projection (+ bias, + activation).  Inside
    tacotron_v1.DecoderV1 the projection runs in the V1 decoder step of the HIP library (csrc/v1.h, v1_launch_step: a dense
    launch on [context | attention cell output]); used on its own, ``call`` runs the wrapped cell and projects its output
    with t2_dense.  ``weights`` = {"kernel": [cell output size, output_size], "bias": [output_size]} (TF names :166-172)."""

    _ACT = {None: 0, "relu": 1, "tanh": 2}

    def __init__(self, cell, output_size, activation=None, use_bias=True, weights=None, name=None, dtype=None):
        if output_size < 1:
            raise ValueError("Parameter output_size must be > 0: %d." % output_size)
        self._cell = cell
        self._output_size = output_size
        self._activation = activation
        self._use_bias = use_bias
        self.weights = weights

    @property
    def output_size(self):
        return self._output_size

    def call(self, inputs, state):
        """rnn_wrappers.py:174-186: output, res_state = cell(inputs, state); tensordot(output, kernel) (+ bias) (activation)."""
        from . import _lib
        from .engine import shared_context
        if self.weights is None:
            raise ValueError("OutputProjectionWrapper called without weights")
        output, res_state = self._cell(inputs, state)
        ctx = shared_context(output.device.index if isinstance(output, torch.Tensor) and output.is_cuda else torch.cuda.current_device())
        x = ctx.f32(output)
        k = ctx.f32(self.weights["kernel"])
        b = ctx.f32(self.weights["bias"]) if self._use_bias else None
        if tuple(k.shape) != (x.shape[-1], self._output_size):
            raise ValueError(f"kernel shape {tuple(k.shape)} does not fit the cell output ({x.shape[-1]}, {self._output_size})")
        fused = self._activation if (self._activation is None or isinstance(self._activation, str)) else None
        if fused not in self._ACT:
            raise ValueError(f"activation {self._activation!r}: None, 'relu', 'tanh' or a callable on tensors")
        rows = int(np.prod(x.shape[:-1]))
        y = torch.empty(tuple(x.shape[:-1]) + (self._output_size,), dtype=torch.float32, device=ctx.device)
        with ctx.ordered():
            _lib.check(ctx.lib.t2_dense(ctx.handle, x.data_ptr(), k.data_ptr(), None if b is None else b.data_ptr(), rows, int(x.shape[-1]),
                                        self._output_size, self._ACT[fused], y.data_ptr()))
        if callable(self._activation):
            y = self._activation(y)
        return y, res_state


class OutputAndStopTokenWrapper(_Delegating):
    """rnn_wrappers.py:81-109.  ``cell`` is a DecoderRNNV2.  Binding (first ``zero_state``) creates the
    HIP model for the whole cell stack; ``call`` is one t2_decode_step and advances ``state`` in place.

    ``weights``: optional dict with the decoder / attention variables under the names of
    tacotron2_amd.weights.weight_shapes; missing variables are initialised as TF would
    (Glorot-uniform kernels, zero biases).  ``n_feed_units`` = num_mels * n_feed_frame, the width of
    the decoder input (defaults to ``out_units``, i.e. n_feed_frame == outputs_per_step)."""

    def __init__(self, cell, out_units, weights: Optional[Dict[str, np.ndarray]] = None, n_feed_units=None, engine=None,
                 seed=1234, num_mels=None, name=None, dtype=None):
        self._out_units = out_units
        self._cell = cell
        self._weights = weights
        self._n_feed_units = n_feed_units or out_units
        self._engine = engine
        self._seed = seed
        self._num_mels = num_mels
        self._step_checked = False
        self.mask_seed = seed          # seed of the sampled masks of single cell calls (``call`` with active dropout / zoneout)

    @property
    def output_size(self):
        return (self._out_units, 1)

    # -- binding to the HIP library ---------------------------------------------------------
    def decoder_dims(self, allow_sampling=False) -> T2Dims:
        """``allow_sampling``: the caller runs the whole teacher-forced loop (DecoderV2.call -> t2_train_decoder_forward),
        where sampled zoneout / PreNet dropout are built; the per-step ``call`` protocol has inference arithmetic only."""
        att = self._cell.attention_cell
        mech = att.mechanism
        g = self._num_mels or gcd(self._out_units, self._n_feed_units)
        lstms = self._cell.lstms
        if lstms[0].num_units != lstms[1].num_units:
            raise ValueError("DecoderRNNV2 uses one out_units for both LSTM layers")
        zc = {att.lstm._zoneout_cell, lstms[0]._zoneout_cell, lstms[1]._zoneout_cell}
        zh = {att.lstm._zoneout_outputs, lstms[0]._zoneout_outputs, lstms[1]._zoneout_outputs}
        if len(zc) != 1 or len(zh) != 1:
            raise NotImplementedError("one pair of zoneout factors per decoder is supported")
        return T2Dims(num_symbols=2, embedding_dim=mech.values_depth, enc_conv_layers=0, enc_units=mech.values_depth,
                      prenet_units=tuple(p.out_units for p in att.prenets), attention_lstm_units=att.lstm.num_units,
                      decoder_lstm_units=lstms[0].num_units, attention_dim=mech.num_units,
                      attention_filters=mech.attention_filters, attention_kernel=mech.attention_kernel,
                      cumulative_weights=mech.cumulative_weights, num_mels=g, outputs_per_step=self._out_units // g,
                      n_feed_frame=self._n_feed_units // g, postnet_layers=1, postnet_kernel=1, postnet_channels=1,
                      zoneout_cell=zc.pop(), zoneout_output=zh.pop())

    def engine(self):
        if self._engine is None:
            from .engine import Engine
            dims = self.decoder_dims(allow_sampling=True)     # sizes only; ``call`` checks the mode again
            w = init_weights(dims, self._seed)
            if self._weights:
                for k, v in self._weights.items():
                    if k in w and k.split('/')[0] in ('decoder', 'attention'):
                        w[k] = np.asarray(v, dtype=np.float32)
            self._engine = Engine(dims, w)
        return self._engine

    def zero_state(self, batch_size, dtype=None):
        mech = self._cell.attention_cell.mechanism
        if batch_size != mech.batch_size:
            raise ValueError("batch_size does not match the attention memory")
        return self.engine().new_state(mech.memory, mech.memory_sequence_length)

    def sampling(self):
        """(PreNet dropout rate, sampled zoneout?) of a single cell call: what the reference's flags switch on.  PreNet dropout
        runs when ``is_training or apply_dropout_on_inference`` (modules.py:96-98), ZoneoutLSTMCell samples when ``is_training``
        (modules.py:274-276).  The library draws one rate for all PreNet layers and one mode for the three cells."""
        att = self._cell.attention_cell
        active = [p for p in att.prenets if p.dropout_enabled and p.drop_rate > 0]
        rates = {float(p.drop_rate) for p in active}
        if active and (len(active) != len(att.prenets) or len(rates) != 1):
            raise NotImplementedError("PreNet dropout: one rate, on all layers or on none")
        cells = (att.lstm,) + tuple(self._cell.lstms)
        zs = [bool(c.is_training and (c._zoneout_cell > 0 or c._zoneout_outputs > 0)) for c in cells]
        if any(zs) and not all(zs):
            raise NotImplementedError("zoneout sampling: all three cells in training mode or none")
        return (rates.pop() if rates else 0.0), all(zs)

    def call(self, inputs, state):
        """One OutputAndStopTokenWrapper.call (rnn_wrappers.py:105-109).  With active PreNet dropout / training-mode zoneout
        (rnn_wrappers.py:51-53, modules.py:96-98,274-276) the step samples its masks from the library's counter-based generator
        at (``mask_seed``, stream, the state's step count, row, unit): the masks DecoderV2.call's fused loop draws at that step."""
        if inputs.shape[-1] != self._n_feed_units:
            raise ValueError(f"decoder input width {inputs.shape[-1]} != n_feed_units {self._n_feed_units}")
        if not self._step_checked:
            self.decoder_dims()
            self._sampling = self.sampling()
            self._step_checked = True
        rate, sample_zoneout = self._sampling
        if rate > 0 or sample_zoneout:
            mel_output, stop_token = self.engine().decode_step_train(real_state(state), inputs, rate, sample_zoneout, self.mask_seed)
        else:
            mel_output, stop_token = self.engine().decode_step(real_state(state), inputs)
        return (mel_output, stop_token), state
