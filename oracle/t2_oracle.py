"""CPU ORACLE -- test infrastructure, NOT product code.

A NumPy fp32 restatement of the Tacotron2 (V2) mel-generation path of
nii-yamagishilab/tacotron2, written from the reference's Python sources (cited per function,
paths relative to /root/reference) and from the TensorFlow r1.11-1.15 op semantics collected
in SURVEY.md Appendix A (TensorFlow itself is not available anywhere in this pipeline).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker.  The product (``tacotron2_amd``) never imports it.

PARITY STATUS: **parity unpinned at the TensorFlow boundary.**  The reference ships no golden
vectors (its only test, tests/decoder_test.py, is a self-consistency property of the V1
decoder) and TensorFlow cannot be run here, so this restatement is pinned only by (a) the
reference property (free-run output fed back as teacher-forcing target reproduces itself),
(b) an independently written torch-CPU implementation of every primitive
(oracle/torch_check.py) and (c) the quirk tests of SURVEY.md Appendix B.

The V2 assembly (``decoder_v2``, ``tacotron2_forward``) does not exist in the reference
(README.md:93 lists it as ToDo); it follows DecoderV1.call (tacotron/tacotron_v1.py:118-153)
and the V1 model_fn (tacotron/models.py:18-129) with the V2 blocks swapped in.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------
# primitives (TF op semantics)
# ----------------------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x, dtype=F32)
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def dense(x, kernel, bias=None, activation=None):
    """tf.layers.Dense: y = x . kernel[in,out] + bias, then activation (App. A3)."""
    y = np.matmul(x.astype(F32), kernel.astype(F32))
    if bias is not None:
        y = y + bias.astype(F32)
    if activation is not None:
        y = activation(y)
    return y.astype(F32)


def relu(x):
    return np.maximum(x, F32(0.0)).astype(F32)


def conv1d_same(x, kernel, bias=None):
    """tf.layers.Conv1D(padding="SAME", strides=1): cross-correlation, kernel [k, in, out],
    pad_left = (k-1)//2, the rest on the right (App. A3).  x: [B, T, in] -> [B, T, out]."""
    k, cin, cout = kernel.shape
    B, T, _ = x.shape
    pad_l = (k - 1) // 2
    pad_r = (k - 1) - pad_l
    xp = np.pad(x.astype(F32), ((0, 0), (pad_l, pad_r), (0, 0)))
    y = np.zeros((B, T, cout), dtype=F32)
    for j in range(k):
        y += np.matmul(xp[:, j:j + T, :], kernel[j].astype(F32))
    if bias is not None:
        y = y + bias.astype(F32)
    return y.astype(F32)


def batch_norm_inference(x, gamma, beta, mean, var, eps=1e-3):
    """tf.layers.batch_normalization(training=False), axis=-1 (App. A4)."""
    inv = (gamma.astype(F32) / np.sqrt(var.astype(F32) + F32(eps))).astype(F32)
    return ((x - mean.astype(F32)) * inv + beta.astype(F32)).astype(F32)


def conv1d_block(x, w: Dict[str, np.ndarray], prefix: str, activation, eps=1e-3):
    """tacotron/modules.py:149-157 (Conv1d.call) with is_training=False: conv1d (no bias) ->
    batch norm (moving statistics) -> activation -> dropout (identity when not training).
    No sequence-length masking anywhere (modules.py:149-157)."""
    y = conv1d_same(x, w[f"{prefix}/kernel"])
    bn = f"{prefix}/batch_normalization"
    y = batch_norm_inference(y, w[f"{bn}/gamma"], w[f"{bn}/beta"], w[f"{bn}/moving_mean"],
                             w[f"{bn}/moving_variance"], eps)
    if activation is not None:
        y = activation(y)
    return y.astype(F32)


def lstm_cell(x, c_prev, h_prev, kernel, bias, forget_bias=1.0):
    """tf.nn.rnn_cell.LSTMCell / LSTMBlockCell (tacotron/rnn_impl.py:25-33; App. A1):
    z = [x, h_prev] . kernel + bias; i, j, f, o = split(z, 4);
    c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(j); h = sigmoid(o) * tanh(c)."""
    z = np.matmul(np.concatenate([x, h_prev], axis=-1).astype(F32), kernel.astype(F32)) + bias.astype(F32)
    i, j, f, o = np.split(z.astype(F32), 4, axis=-1)
    c = sigmoid(f + F32(forget_bias)) * c_prev + sigmoid(i) * np.tanh(j)
    h = sigmoid(o) * np.tanh(c)
    return c.astype(F32), h.astype(F32)


def zoneout_lstm_cell(x, state, kernel, bias, zoneout_cell, zoneout_output, forget_bias=1.0):
    """tacotron/modules.py:261-283 (ZoneoutLSTMCell.__call__), inference branch (:277-279):
    state c/h are blended with the previous state, but the *output* is the raw new h (:263,283)."""
    c_prev, h_prev = state
    new_c, new_h = lstm_cell(x, c_prev, h_prev, kernel, bias, forget_bias)
    zc, zh = F32(zoneout_cell), F32(zoneout_output)
    c = (F32(1.0) - zc) * new_c + zc * c_prev
    h = (F32(1.0) - zh) * new_h + zh * h_prev
    return new_h, (c.astype(F32), h.astype(F32))


# ----------------------------------------------------------------------------------------
# training-mode sampling.  TensorFlow's random ops cannot be reproduced (stateful Philox streams keyed by
# graph-level seeds), so "parity" for the sampled paths means: the SAME counter-based generator in the
# oracle and in the HIP kernels (tacotron2_amd/csrc/kernels.h: t2_u01), TF's arithmetic around it.
# ----------------------------------------------------------------------------------------
def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def rng_u01(seed, stream, t, rows, cols):
    """Uniform [0,1) field of shape [rows, cols] for (seed, stream, t): 24 random bits per element."""
    r = np.arange(rows, dtype=np.uint64)[:, None]
    c = np.arange(cols, dtype=np.uint64)[None, :]
    h = _mix32(np.uint64(seed & 0xFFFFFFFF) ^ ((np.uint64(stream) * 0x9E3779B9) & 0xFFFFFFFF))
    h = _mix32(h ^ ((np.uint64(t) * 0x85EBCA6B + 0x165667B1) & 0xFFFFFFFF))
    h = _mix32(h ^ ((r * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF))
    h = _mix32(h ^ ((c * 0x9E3779B1 + 0x85EBCA77) & 0xFFFFFFFF))
    return ((h >> 8).astype(np.float64) * (1.0 / 16777216.0)).astype(F32)


def dropout_train(x, rate, u):
    """tf.layers.dropout(x, rate, training=True) = tf.nn.dropout(x, keep_prob = 1 - rate) of TF 1.11/1.12:
    x / keep_prob * floor(keep_prob + uniform); u is the uniform field (keep <=> u >= rate)."""
    keep = F32(1.0 - rate)
    mask = (u >= F32(rate)).astype(F32)
    return ((x.astype(F32) / keep) * mask).astype(F32)


def zoneout_lstm_cell_train(x, state, kernel, bias, zoneout_cell, zoneout_output, u_c, u_h, forget_bias=1.0):
    """ZoneoutLSTMCell.__call__, training branch (modules.py:272-276):
    c = keep_c * dropout(new_c - prev_c, keep_prob=keep_c) + prev_c (same for h); output is the raw new h."""
    c_prev, h_prev = state
    new_c, new_h = lstm_cell(x, c_prev, h_prev, kernel, bias, forget_bias)
    kc, kh = F32(1.0 - zoneout_cell), F32(1.0 - zoneout_output)
    c = kc * dropout_train(new_c - c_prev, zoneout_cell, u_c) + c_prev
    h = kh * dropout_train(new_h - h_prev, zoneout_output, u_h) + h_prev
    return new_h, (c.astype(F32), h.astype(F32))


# stream ids of the generator (shared with csrc/t2mi.hip)
STREAM_PRENET = 1          # + layer index
STREAM_ZONE_ATT, STREAM_ZONE_L1, STREAM_ZONE_L2 = 10, 12, 14      # cell mask; + 1: output mask


def sequence_mask(lengths, maxlen):
    return np.arange(maxlen)[None, :] < np.asarray(lengths)[:, None]


def reverse_sequence(x, lengths):
    """tf.reverse_sequence(x, lengths, seq_axis=1, batch_axis=0)."""
    y = x.copy()
    for b, n in enumerate(np.asarray(lengths)):
        n = int(n)
        y[b, :n] = x[b, :n][::-1]
    return y


def dynamic_rnn_zoneout_lstm(x, lengths, kernel, bias, units, zc, zh, forget_bias=1.0):
    """tf.nn.dynamic_rnn over a ZoneoutLSTMCell with sequence_length (App. A8): for rows with
    t >= len the output is zero and the state is copied through."""
    B, T, _ = x.shape
    c = np.zeros((B, units), dtype=F32)
    h = np.zeros((B, units), dtype=F32)
    out = np.zeros((B, T, units), dtype=F32)
    for t in range(T):
        o, (c2, h2) = zoneout_lstm_cell(x[:, t], (c, h), kernel, bias, zc, zh, forget_bias)
        if lengths is None:
            c, h, out[:, t] = c2, h2, o
        else:
            live = (t < np.asarray(lengths))[:, None]
            c = np.where(live, c2, c).astype(F32)
            h = np.where(live, h2, h).astype(F32)
            out[:, t] = np.where(live, o, F32(0.0))
    return out


def embedding_lookup(ids, table, index_offset=0):
    """tacotron/modules.py:64-67 (Embedding.call): range assertion then table lookup."""
    ids = np.asarray(ids)
    if np.any(ids < index_offset) or np.any(ids >= index_offset + table.shape[0]):
        raise AssertionError("embedding id out of range")  # tf.assert_greater_equal / assert_less
    return table[ids - index_offset].astype(F32)


# ----------------------------------------------------------------------------------------
# encoder / postnet
# ----------------------------------------------------------------------------------------
def encoder_v2(d, w, emb_out, input_lengths=None):
    """tacotron/tacotron_v2.py:53-63 (EncoderV2.call): conv stack (relu), then
    bidirectional_dynamic_rnn of two ZoneoutLSTMCell(out_units // 2), outputs concatenated."""
    x = emb_out
    for i in range(d.enc_conv_layers):
        x = conv1d_block(x, w, f"encoder/conv1d_{i}", relu, d.bn_eps)
    h = d.enc_units // 2
    B, T, _ = x.shape
    p = "encoder/bidirectional_rnn"
    fw = dynamic_rnn_zoneout_lstm(x, input_lengths, w[f"{p}/fw/lstm_cell/kernel"], w[f"{p}/fw/lstm_cell/bias"],
                                  h, d.zoneout_cell, d.zoneout_output, d.forget_bias)
    lens = input_lengths if input_lengths is not None else np.full((B,), T)
    xr = reverse_sequence(x, lens)
    bw = dynamic_rnn_zoneout_lstm(xr, input_lengths, w[f"{p}/bw/lstm_cell/kernel"], w[f"{p}/bw/lstm_cell/bias"],
                                  h, d.zoneout_cell, d.zoneout_output, d.forget_bias)
    bw = reverse_sequence(bw, lens)
    return np.concatenate([fw, bw], axis=-1).astype(F32)


def postnet_v2(d, w, mel):
    """tacotron/tacotron_v2.py:200-204 (PostNetV2.call): (L-1) x Conv1d(tanh) + Conv1d(None),
    all with out_channels, then Dense(out_units) and the residual on the raw input."""
    x = mel.astype(F32)
    for i in range(1, d.postnet_layers + 1):
        act = np.tanh if i < d.postnet_layers else None
        x = conv1d_block(x, w, f"postnet/conv1d_{i}", act, d.bn_eps)
    projected = dense(x, w["postnet/projection/kernel"], w["postnet/projection/bias"])
    return (mel.astype(F32) + projected).astype(F32)


# ----------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------
class LocationSensitiveAttention:
    """tacotron/tacotron_v2.py:81-144 over tf.contrib.seq2seq.BahdanauAttention (App. A5)."""

    def __init__(self, d, w, memory, memory_sequence_length, smoothing=False):
        if smoothing:
            # tacotron_v2.py:93,143-144: _smoothing_normalization lacks ``self`` -> TypeError in
            # the reference as soon as it is called; not part of the path.
            raise TypeError("_smoothing_normalization() takes 1 positional argument but 3 were given")
        B, T, _ = memory.shape
        self.T = T
        self.lengths = None if memory_sequence_length is None else np.asarray(memory_sequence_length)
        # _prepare_memory: values = memory * sequence_mask
        if self.lengths is not None:
            self.values = (memory * sequence_mask(self.lengths, T)[:, :, None]).astype(F32)
        else:
            self.values = memory.astype(F32)
        self.keys = dense(self.values, w["attention/memory_layer/kernel"])        # no bias
        self.wq = w["attention/query_layer/kernel"]
        self.conv_k = w["attention/location_features_convolution/kernel"]
        self.conv_b = w["attention/location_features_convolution/bias"]
        self.wloc = w["attention/location_features_layer/kernel"]
        self.v_a = w["attention/attention_variable"]
        self.b_a = w["attention/attention_bias"]
        self.cumulative = bool(d.cumulative_weights)

    def initial_state(self, B):
        return np.zeros((B, self.T), dtype=F32)

    def __call__(self, query, state):
        """tacotron_v2.py:118-141."""
        previous_alignments = state
        processed_query = dense(query, self.wq)[:, None, :]                          # :122-125
        f = conv1d_same(previous_alignments[:, :, None], self.conv_k, self.conv_b)   # :128-130
        processed_location = dense(f, self.wloc)                                     # :131
        # _location_sensitive_score :66-78
        energy = np.sum(self.v_a.astype(F32) * np.tanh(self.keys + processed_query + processed_location
                                                       + self.b_a.astype(F32)), axis=2).astype(F32)
        # _maybe_mask_score(-inf) + softmax  (App. A5)
        if self.lengths is not None:
            energy = np.where(sequence_mask(self.lengths, self.T), energy, F32(-np.inf)).astype(F32)
        m = np.max(energy, axis=-1, keepdims=True)
        e = np.exp(energy - m)
        alignments = (e / np.sum(e, axis=-1, keepdims=True)).astype(F32)
        next_state = alignments + previous_alignments if self.cumulative else alignments  # :137-140
        return alignments, next_state.astype(F32)


# ----------------------------------------------------------------------------------------
# decoder cell stack
# ----------------------------------------------------------------------------------------
@dataclass
class AttentionWrapperState:
    """tf.contrib.seq2seq.AttentionWrapperState (App. A6)."""
    cell_state: Tuple[np.ndarray, np.ndarray]
    attention: np.ndarray
    time: int
    alignments: np.ndarray
    alignment_history: List[np.ndarray]
    attention_state: np.ndarray


class DecoderCellV2:
    """OutputAndStopTokenWrapper(DecoderRNNV2(AttentionRNN(ZoneoutLSTM, prenets, LSA))).

    tacotron/rnn_wrappers.py:81-109 (mel / stop projections), tacotron/tacotron_v2.py:147-178
    (MultiRNNCell of attention cell + 2 ZoneoutLSTM, no residuals), rnn_wrappers.py:112-144
    (AttentionWrapper(alignment_history=True, output_attention=False) -> prenet -> concat)."""

    def __init__(self, d, w, memory, memory_sequence_length, train_seed=None, prenet_drop_rate=0.5):
        """train_seed: None = inference arithmetic; an int = training arithmetic (PreNet dropout with
        prenet_drop_rate, zoneout sampling) drawn from rng_u01(train_seed, stream, time, ...)."""
        self.d, self.w = d, w
        self.mech = LocationSensitiveAttention(d, w, memory, memory_sequence_length)
        self.B = memory.shape[0]
        self.E = memory.shape[2]
        self.train_seed, self.prenet_drop_rate = train_seed, prenet_drop_rate

    def _lstm(self, x, state, name, units, stream, t):
        d, w = self.d, self.w
        if self.train_seed is None:
            return zoneout_lstm_cell(x, state, w[f"{name}/kernel"], w[f"{name}/bias"], d.zoneout_cell, d.zoneout_output, d.forget_bias)
        u_c = rng_u01(self.train_seed, stream, t, self.B, units)
        u_h = rng_u01(self.train_seed, stream + 1, t, self.B, units)
        return zoneout_lstm_cell_train(x, state, w[f"{name}/kernel"], w[f"{name}/bias"], d.zoneout_cell, d.zoneout_output,
                                       u_c, u_h, d.forget_bias)

    def zero_state(self):
        d, B = self.d, self.B
        z = lambda n: np.zeros((B, n), dtype=F32)
        att = AttentionWrapperState(cell_state=(z(d.attention_lstm_units), z(d.attention_lstm_units)),
                                    attention=z(self.E), time=0, alignments=self.mech.initial_state(B),
                                    alignment_history=[], attention_state=self.mech.initial_state(B))
        return (att, (z(d.decoder_lstm_units), z(d.decoder_lstm_units)),
                (z(d.decoder_lstm_units), z(d.decoder_lstm_units)))

    def __call__(self, inputs, state):
        d, w = self.d, self.w
        att_state, s1, s2 = state
        # DecoderPreNetWrapper (rnn_wrappers.py:51-53); PreNet.call modules.py:88-91, dropout off
        x = inputs.astype(F32)
        t = att_state.time
        for i in range(len(d.prenet_units)):
            x = dense(x, w[f"decoder/prenet_{i}/dense/kernel"], w[f"decoder/prenet_{i}/dense/bias"], relu)
            if self.train_seed is not None:                      # modules.py:90,96-98
                x = dropout_train(x, self.prenet_drop_rate, rng_u01(self.train_seed, STREAM_PRENET + i, t, self.B, x.shape[1]))
        # AttentionWrapper.call (App. A6)
        cell_inputs = np.concatenate([x, att_state.attention], axis=-1)
        cell_output, next_cell_state = self._lstm(cell_inputs, att_state.cell_state, "decoder/attention_lstm",
                                                  d.attention_lstm_units, STREAM_ZONE_ATT, t)
        alignments, next_attention_state = self.mech(cell_output, att_state.attention_state)
        context = np.matmul(alignments[:, None, :], self.mech.values)[:, 0, :].astype(F32)
        new_att = AttentionWrapperState(cell_state=next_cell_state, attention=context, time=att_state.time + 1,
                                        alignments=alignments,
                                        alignment_history=att_state.alignment_history + [alignments],
                                        attention_state=next_attention_state)
        # ConcatOutputAndAttentionWrapper (rnn_wrappers.py:76-78): the NEW context
        y0 = np.concatenate([cell_output, context], axis=-1)
        # DecoderRNNV2 / MultiRNNCell (tacotron_v2.py:155-161)
        y1, ns1 = self._lstm(y0, s1, "decoder/lstm_1", d.decoder_lstm_units, STREAM_ZONE_L1, t)
        y2, ns2 = self._lstm(y1, s2, "decoder/lstm_2", d.decoder_lstm_units, STREAM_ZONE_L2, t)
        # OutputAndStopTokenWrapper.call (rnn_wrappers.py:105-109)
        mel = dense(y2, w["decoder/out_projection/kernel"], w["decoder/out_projection/bias"])
        stop = dense(y2, w["decoder/stop_token_projection/kernel"], w["decoder/stop_token_projection/bias"])
        return (mel, stop), (new_att, ns1, ns2)


# ----------------------------------------------------------------------------------------
# helpers (tacotron/helpers.py)
# ----------------------------------------------------------------------------------------
def _go_frames(batch_size, output_dim):
    """helpers.py:187-188."""
    return np.zeros((batch_size, output_dim), dtype=F32)


class InferenceHelper:
    """helpers.py:13-49 (legacy; finishes a row when its whole output equals 0.0)."""

    def __init__(self, batch_size, output_dim, r, n_feed_frame=1):
        assert n_feed_frame <= r
        self.batch_size, self.output_dim, self.r, self.n_feed_frame = batch_size, output_dim, r, n_feed_frame

    def initialize(self):
        return np.zeros((self.batch_size,), dtype=bool), _go_frames(self.batch_size, self.output_dim * self.n_feed_frame)

    def next_inputs(self, time, outputs, state):
        out, _done = outputs
        finished = np.all(out == F32(0.0), axis=1)
        return finished, out[:, -self.output_dim * self.n_feed_frame:], state


class StopTokenBasedInferenceHelper:
    """helpers.py:52-94."""

    def __init__(self, batch_size, output_dim, r, n_feed_frame=1, min_iters=10):
        assert n_feed_frame <= r
        self.batch_size, self.output_dim, self.n_feed_frame, self.min_iters = batch_size, output_dim, n_feed_frame, min_iters

    def initialize(self):
        return np.zeros((self.batch_size,), dtype=bool), _go_frames(self.batch_size, self.output_dim * self.n_feed_frame)

    def is_finished(self, done, time):
        termination_criteria = sigmoid(done) > F32(0.5)           # [B, 1]
        minimum_requirement = time > self.min_iters
        termination = np.logical_and(termination_criteria, minimum_requirement)
        return np.all(termination, axis=0)                        # shape [1]: batch-coupled

    def next_inputs(self, time, outputs, state):
        out, done = outputs
        finished = self.is_finished(done, time)
        return finished, out[:, -self.output_dim * self.n_feed_frame:], state


class ValidationHelper:
    """helpers.py:97-138."""

    def __init__(self, targets, batch_size, output_dim, r, n_feed_frame=1, teacher_forcing=False):
        assert n_feed_frame <= r
        self.batch_size, self.output_dim, self.n_feed_frame = batch_size, output_dim, n_feed_frame
        self.num_steps = targets.shape[1] // r
        self.teacher_forcing = teacher_forcing
        self._targets = targets.reshape(batch_size, self.num_steps, output_dim * r)

    def initialize(self):
        return np.zeros((self.batch_size,), dtype=bool), _go_frames(self.batch_size, self.output_dim * self.n_feed_frame)

    def next_inputs(self, time, outputs, state):
        out, _done = outputs
        finished = np.full((self.batch_size,), time + 1 >= self.num_steps)
        n = self.output_dim * self.n_feed_frame
        nxt = self._targets[:, time, -n:] if self.teacher_forcing else out[:, -n:]
        return finished, nxt, state


class TrainingHelper:
    """helpers.py:141-184 (full length for every row: padding frames are not masked)."""

    def __init__(self, targets, output_dim, r, n_feed_frame=1):
        assert n_feed_frame <= r
        B, T = targets.shape[0], targets.shape[1]
        self.batch_size, self.output_dim, self.n_feed_frame = B, output_dim, n_feed_frame
        self._targets = targets.reshape(B, T // r, output_dim * r)
        self._lengths = np.full((B,), self._targets.shape[1])

    def initialize(self):
        return np.zeros((self.batch_size,), dtype=bool), _go_frames(self.batch_size, self.output_dim * self.n_feed_frame)

    def next_inputs(self, time, outputs, state):
        finished = (time + 1 >= self._lengths)
        n = self.output_dim * self.n_feed_frame
        return finished, self._targets[:, time, -n:], state


def dynamic_decode(cell, helper, initial_state, maximum_iterations):
    """tf.contrib.seq2seq.dynamic_decode(BasicDecoder(cell, helper, state), maximum_iterations)
    with impute_finished=False (App. A7; call site tacotron/tacotron_v1.py:149-150)."""
    finished, inputs = helper.initialize()
    if maximum_iterations is not None:
        finished = np.logical_or(finished, 0 >= maximum_iterations)
    state = initial_state
    time = 0
    mels, stops = [], []
    while not np.all(finished):
        outputs, state = cell(inputs, state)
        step_finished, inputs, state = helper.next_inputs(time, outputs, state)
        next_finished = np.logical_or(step_finished, finished)        # broadcasts the [1] case
        if maximum_iterations is not None:
            next_finished = np.logical_or(next_finished, time + 1 >= maximum_iterations)
        mels.append(outputs[0])
        stops.append(outputs[1])
        finished = next_finished
        time += 1
    B = helper.batch_size
    if time == 0:
        return (np.zeros((B, 0, 0), F32), np.zeros((B, 0, 1), F32)), state, 0
    return (np.stack(mels, axis=1), np.stack(stops, axis=1)), state, time


def decoder_v2(d, w, memory, is_training=False, is_validation=False, teacher_forcing=False,
               memory_sequence_length=None, target=None, max_iters=None, train_seed=None, prenet_drop_rate=0.5):
    """V2 counterpart of DecoderV1.call (tacotron/tacotron_v1.py:118-153): same helper selection
    (:134-147), same dynamic_decode call (:149-150), same reshape (:152).
    Arithmetic is the inference-mode arithmetic (dropout off, deterministic zoneout) unless train_seed is
    given: then PreNet dropout and zoneout are sampled (training-mode forward)."""
    B = memory.shape[0]
    cell = DecoderCellV2(d, w, memory, memory_sequence_length, train_seed=train_seed, prenet_drop_rate=prenet_drop_rate)
    M, r, nf = d.num_mels, d.outputs_per_step, d.n_feed_frame
    if is_training:
        helper = TrainingHelper(target, M, r, nf)
    elif is_validation:
        helper = ValidationHelper(target, B, M, r, nf, teacher_forcing=teacher_forcing)
    else:
        helper = StopTokenBasedInferenceHelper(B, M, r, nf, min_iters=d.min_iters)
    (dec_out, stop), final_state, steps = dynamic_decode(
        cell, helper, cell.zero_state(), d.max_iters if max_iters is None else max_iters)
    mel = dec_out.reshape(B, -1, M)
    return mel, stop, final_state


def alignment_from_state(final_state):
    """models.py:53: transpose(alignment_history.stack(), [1, 2, 0]) -> [B, T_in, S]."""
    hist = np.stack(final_state[0].alignment_history, axis=0)   # [S, B, T_in]
    return np.transpose(hist, (1, 2, 0))


def tacotron2_forward(d, w, ids, source_length, mode="predict", target=None, teacher_forcing=False,
                      max_iters=None):
    """V2 counterpart of the V1 model_fn forward (tacotron/models.py:44-53, 122-129):
    embedding -> EncoderV2 -> decoder -> PostNetV2; returns dict(mel, mel_postnet, stop, alignment)."""
    emb = embedding_lookup(ids, w["embedding"])
    memory = encoder_v2(d, w, emb, source_length)
    mel, stop, st = decoder_v2(d, w, memory, is_training=(mode == "train"), is_validation=(mode == "eval"),
                               teacher_forcing=teacher_forcing, memory_sequence_length=source_length,
                               target=target, max_iters=max_iters)
    return {"memory": memory, "mel": mel, "stop": stop, "alignment": alignment_from_state(st),
            "mel_postnet": postnet_v2(d, w, mel), "steps": stop.shape[1]}


# ----------------------------------------------------------------------------------------
# training-step scalar math (models.py:136-154)
# ----------------------------------------------------------------------------------------
def spec_loss(y_hat, y, mask):
    """models.py:136-144 with tf.losses.compute_weighted_loss (SUM_BY_NONZERO_WEIGHTS, App. A9)."""
    l1 = np.abs(y_hat - y).astype(F32)
    wgt = np.broadcast_to(mask[:, :, None].astype(F32), l1.shape)
    nz = np.count_nonzero(wgt)
    return F32(np.sum(l1 * wgt, dtype=np.float64) / nz) if nz else F32(0.0)


def binary_loss(done_hat, done, mask):
    """models.py:146-148: sigmoid cross entropy, weights = mask, SUM_BY_NONZERO_WEIGHTS."""
    x = done_hat[..., 0].astype(F32)
    z = done.astype(F32)
    ce = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    nz = np.count_nonzero(mask)
    return F32(np.sum(ce * mask.astype(F32), dtype=np.float64) / nz) if nz else F32(0.0)


def learning_rate_decay(init_rate, global_step):
    """models.py:150-154."""
    warmup_steps = 4000.0
    step = float(global_step + 1)
    return init_rate * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


def loss_parts(l, wgt):
    """(sum of weighted losses, number of non-zero broadcast weights): the two numbers a data-parallel
    job all-reduces before dividing."""
    wgt = np.broadcast_to(wgt.astype(F32), l.shape)
    return float(np.sum(l.astype(F32) * wgt, dtype=np.float64)), int(np.count_nonzero(wgt))


def global_norm(grads):
    """tf.global_norm: sqrt(sum over all tensors of sum(g^2))."""
    return float(np.sqrt(sum(float(np.sum(np.asarray(g, np.float64) ** 2)) for g in grads)))


def clip_by_global_norm(grads, clip_norm):
    """tf.clip_by_global_norm (models.py:71): g * clip_norm / max(global_norm, clip_norm)."""
    norm = global_norm(grads)
    scale = clip_norm / max(norm, clip_norm)
    return [(np.asarray(g, F32) * F32(scale)).astype(F32) for g in grads], norm


def adam_apply(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer._apply_dense at step t (1-based), fp32 slots (models.py:67-68,76; App. A of
    SURVEY.md): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps) -- epsilon outside
    the square root, unlike torch.optim.Adam."""
    lr_t = F32(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    b1, b2, e = F32(beta1), F32(beta2), F32(eps)
    g = np.asarray(g, F32)
    m = (b1 * m + (F32(1) - b1) * g).astype(F32)
    v = (b2 * v + (F32(1) - b2) * g * g).astype(F32)
    p = (p - lr_t * m / (np.sqrt(v) + e)).astype(F32)
    return p, m, v
