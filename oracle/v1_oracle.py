"""CPU ORACLE (Tacotron v1 path) -- test infrastructure, NOT product code.

NumPy fp32 restatement of the reference's V1 blocks: CBHG / HighwayNet (tacotron/modules.py:101-232),
EncoderV1, AttentionRNNV1, DecoderRNNV1, DecoderV1, PostNet (tacotron/tacotron_v1.py) and
OutputProjectionWrapper (tacotron/rnn_wrappers.py:147-186), with the TensorFlow op semantics of
SURVEY.md Appendix A (GRUCell A2, MaxPooling1D A3, BahdanauAttention A5, AttentionWrapper A6).

PARITY STATUS: **parity unpinned at the TensorFlow boundary** (same reasons as oracle/t2_oracle.py);
pinned by the reference's own property test (tests/decoder_test.py:42-67, which IS a V1 test) and by the
independent torch-CPU primitives in oracle/torch_check.py.  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np

from .t2_oracle import (F32, AttentionWrapperState, StopTokenBasedInferenceHelper, TrainingHelper, ValidationHelper,
                        conv1d_block, dense, dynamic_decode, embedding_lookup, relu, reverse_sequence, sequence_mask, sigmoid)


def gru_cell(x, h, wg, bg, wc, bc):
    """tf.nn.rnn_cell.GRUCell (rnn_impl.py:38; App. A2): the reset gate multiplies h BEFORE the candidate
    matmul.  Returns the new state (= output)."""
    g = sigmoid(np.matmul(np.concatenate([x, h], axis=-1).astype(F32), wg.astype(F32)) + bg.astype(F32))
    r, u = np.split(g, 2, axis=-1)
    c = np.tanh(np.matmul(np.concatenate([x, r * h], axis=-1).astype(F32), wc.astype(F32)) + bc.astype(F32)).astype(F32)
    return (u * h + (F32(1.0) - u) * c).astype(F32)


def _gru_w(w, p):
    return w[f"{p}/gates/kernel"], w[f"{p}/gates/bias"], w[f"{p}/candidate/kernel"], w[f"{p}/candidate/bias"]


def dynamic_rnn_gru(x, lengths, w, p, units):
    """tf.nn.dynamic_rnn over a GRUCell (App. A8): zero output and state copied through for t >= len."""
    B, T, _ = x.shape
    h = np.zeros((B, units), F32)
    out = np.zeros((B, T, units), F32)
    for t in range(T):
        h2 = gru_cell(x[:, t], h, *_gru_w(w, p))
        if lengths is None:
            h, out[:, t] = h2, h2
        else:
            live = (t < np.asarray(lengths))[:, None]
            h = np.where(live, h2, h).astype(F32)
            out[:, t] = np.where(live, h2, F32(0.0))
    return out


def maxpool2_same(x):
    """tf.layers.MaxPooling1D(pool_size=2, strides=1, padding="SAME") (modules.py:182; App. A3)."""
    y = x.copy()
    y[:, :-1] = np.maximum(x[:, :-1], x[:, 1:])
    return y


def highway(x, w, p):
    """HighwayNet.call (modules.py:124-127)."""
    h = dense(x, w[f"{p}/H/kernel"], w[f"{p}/H/bias"], relu)
    t = dense(x, w[f"{p}/T/kernel"], w[f"{p}/T/bias"], sigmoid)
    return (h * t + x * (F32(1.0) - t)).astype(F32)


def cbhg(w, p, x, out_units, max_filter_width, num_highway, input_lengths=None, eps=1e-3):
    """CBHG.call (modules.py:205-229)."""
    bank = np.concatenate([conv1d_block(x, w, f"{p}/conv1d_K{k}", relu, eps) for k in range(1, max_filter_width + 1)], axis=-1)
    pooled = maxpool2_same(bank)
    p1 = conv1d_block(pooled, w, f"{p}/proj1", relu, eps)
    p2 = conv1d_block(p1, w, f"{p}/proj2", None, eps)              # tf.identity
    hw = (p2 + x).astype(F32)
    half = out_units // 2
    if hw.shape[2] != half:
        hw = dense(hw, w[f"{p}/adjustment/kernel"], w[f"{p}/adjustment/bias"])
    for i in range(1, num_highway + 1):
        hw = highway(hw, w, f"{p}/highway_{i}")
    B, T, _ = hw.shape
    rp = f"{p}/bidirectional_rnn"
    fw = dynamic_rnn_gru(hw, input_lengths, w, f"{rp}/fw/gru_cell", half)
    lens = input_lengths if input_lengths is not None else np.full((B,), T)
    bw = reverse_sequence(dynamic_rnn_gru(reverse_sequence(hw, lens), input_lengths, w, f"{rp}/bw/gru_cell", half), lens)
    return np.concatenate([fw, bw], axis=-1).astype(F32)


def encoder_v1(d, w, emb_out, input_lengths=None):
    """EncoderV1.call (tacotron_v1.py:46-49): PreNets (dropout off) then CBHG."""
    x = emb_out.astype(F32)
    for i in range(len(d.encoder_prenet_out_units)):
        x = dense(x, w[f"v1/encoder/prenet_{i}/dense/kernel"], w[f"v1/encoder/prenet_{i}/dense/bias"], relu)
    return cbhg(w, "v1/encoder/cbhg", x, d.cbhg_out_units, d.max_filter_width, d.num_highway, input_lengths, d.bn_eps)


def postnet_v1(d, w, mel):
    """PostNet.call (tacotron_v1.py:180-183): CBHG (no lengths) then Dense(num_freq)."""
    y = cbhg(w, "v1/postnet/cbhg", mel.astype(F32), d.post_net_cbhg_out_units, d.post_net_max_filter_width,
             d.post_net_num_highway, None, d.bn_eps)
    return dense(y, w["v1/postnet/linear/kernel"], w["v1/postnet/linear/bias"])


class BahdanauAttention:
    """tf.contrib.seq2seq.BahdanauAttention(num_units, memory, memory_sequence_length), normalize=False
    (tacotron_v1.py:59; App. A5)."""

    def __init__(self, w, memory, memory_sequence_length):
        B, T, _ = memory.shape
        self.T = T
        self.lengths = None if memory_sequence_length is None else np.asarray(memory_sequence_length)
        self.values = (memory * sequence_mask(self.lengths, T)[:, :, None]).astype(F32) if self.lengths is not None \
            else memory.astype(F32)
        self.keys = dense(self.values, w["v1/attention/memory_layer/kernel"])
        self.wq, self.v = w["v1/attention/query_layer/kernel"], w["v1/attention/attention_v"]

    def initial_state(self, B):
        return np.zeros((B, self.T), F32)

    def __call__(self, query, state):
        q = dense(query, self.wq)[:, None, :]
        score = np.sum(self.v.astype(F32) * np.tanh(self.keys + q), axis=2).astype(F32)
        if self.lengths is not None:
            score = np.where(sequence_mask(self.lengths, self.T), score, F32(-np.inf)).astype(F32)
        m = np.max(score, axis=-1, keepdims=True)
        e = np.exp(score - m)
        a = (e / np.sum(e, axis=-1, keepdims=True)).astype(F32)
        return a, a


class DecoderCellV1:
    """OutputAndStopTokenWrapper(DecoderRNNV1(OutputProjectionWrapper(AttentionRNN(GRU, prenets, Bahdanau)),
    2 x ResidualWrapper(GRU))) (tacotron_v1.py:56-91,126-130; rnn_wrappers.py:81-186)."""

    def __init__(self, d, w, memory, memory_sequence_length):
        self.d, self.w = d, w
        self.mech = BahdanauAttention(w, memory, memory_sequence_length)
        self.B, self.E = memory.shape[0], memory.shape[2]

    def zero_state(self):
        d, B = self.d, self.B
        z = lambda n: np.zeros((B, n), F32)
        att = AttentionWrapperState(cell_state=z(d.attention_out_units), attention=z(self.E), time=0,
                                    alignments=self.mech.initial_state(B), alignment_history=[],
                                    attention_state=self.mech.initial_state(B))
        return (att, z(d.decoder_out_units), z(d.decoder_out_units))

    def __call__(self, inputs, state):
        d, w = self.d, self.w
        att, h1, h2 = state
        x = inputs.astype(F32)
        for i in range(len(d.decoder_prenet_out_units)):
            x = dense(x, w[f"v1/decoder/prenet_{i}/dense/kernel"], w[f"v1/decoder/prenet_{i}/dense/bias"], relu)
        cell_in = np.concatenate([x, att.attention], axis=-1)                           # AttentionWrapper (A6)
        h_att = gru_cell(cell_in, att.cell_state, *_gru_w(w, "v1/decoder/attention_gru"))
        alignments, next_att_state = self.mech(h_att, att.attention_state)
        context = np.matmul(alignments[:, None, :], self.mech.values)[:, 0, :].astype(F32)
        new_att = AttentionWrapperState(cell_state=h_att, attention=context, time=att.time + 1, alignments=alignments,
                                        alignment_history=att.alignment_history + [alignments],
                                        attention_state=next_att_state)
        y0 = np.concatenate([h_att, context], axis=-1)                                  # rnn_wrappers.py:76-78
        y0 = dense(y0, w["v1/decoder/output_projection/kernel"], w["v1/decoder/output_projection/bias"])   # :180-186
        nh1 = gru_cell(y0, h1, *_gru_w(w, "v1/decoder/gru_1"))
        y1 = (y0 + nh1).astype(F32)                                                     # ResidualWrapper
        nh2 = gru_cell(y1, h2, *_gru_w(w, "v1/decoder/gru_2"))
        y2 = (y1 + nh2).astype(F32)
        mel = dense(y2, w["v1/decoder/out_projection/kernel"], w["v1/decoder/out_projection/bias"])
        stop = dense(y2, w["v1/decoder/stop_token_projection/kernel"], w["v1/decoder/stop_token_projection/bias"])
        return (mel, stop), (new_att, nh1, nh2)


def decoder_v1(d, w, memory, is_training=False, is_validation=False, teacher_forcing=False,
               memory_sequence_length=None, target=None, max_iters=None):
    """DecoderV1.call (tacotron_v1.py:118-153), inference arithmetic (dropout off)."""
    B = memory.shape[0]
    cell = DecoderCellV1(d, w, memory, memory_sequence_length)
    M, r, nf = d.num_mels, d.outputs_per_step, d.n_feed_frame
    if is_training:
        helper = TrainingHelper(target, M, r, nf)
    elif is_validation:
        helper = ValidationHelper(target, B, M, r, nf, teacher_forcing=teacher_forcing)
    else:
        helper = StopTokenBasedInferenceHelper(B, M, r, nf, min_iters=d.min_iters)
    (dec_out, stop), final_state, steps = dynamic_decode(cell, helper, cell.zero_state(),
                                                         d.max_iters if max_iters is None else max_iters)
    return dec_out.reshape(B, -1, M), stop, final_state


def tacotron1_forward(d, w, ids, source_length, mode="predict", target=None, teacher_forcing=False, max_iters=None):
    """V1 model_fn forward (models.py:44-53): the encoder is called WITHOUT lengths (:47), the decoder's
    attention with memory_sequence_length (:51)."""
    emb = embedding_lookup(ids, w["embedding"])
    memory = encoder_v1(d, w, emb, None)
    mel, stop, st = decoder_v1(d, w, memory, is_training=(mode == "train"), is_validation=(mode == "eval"),
                               teacher_forcing=teacher_forcing, memory_sequence_length=source_length, target=target,
                               max_iters=max_iters)
    hist = np.stack(st[0].alignment_history, axis=0)
    return {"memory": memory, "mel": mel, "stop": stop, "alignment": np.transpose(hist, (1, 2, 0)), "steps": stop.shape[1]}
