"""CPU ORACLE for the training step -- test infrastructure, NOT product code.

A differentiable torch (float64, CPU) restatement of the training-mode forward of the V2 path -- Embedding,
EncoderV2 (batch-statistics batch norm, conv dropout, zoneout-sampled BiLSTM), the teacher-forced decoder
(TrainingHelper, PreNet dropout, zoneout-sampled LSTMs, location-sensitive attention) and the two losses of
models.py:57-61 -- whose gradients come from torch.autograd.  It is the gradient checker for the hand-written
HIP backward pass; the sampled masks come from the same counter-based generator as the kernels
(oracle/t2_oracle.py:rng_u01).  PostNetV2 does not enter the loss of the reference's model_fn
(models.py:58-61 uses mel_output and stop_token only), so it has no gradient here.

PARITY STATUS: parity unpinned at the TensorFlow boundary (see oracle/t2_oracle.py); forward values are checked
against the NumPy oracle, gradients additionally by finite differences (tests/test_oracle_train.py)."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import t2_oracle as O

DT = torch.float64


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=DT)


def _u(seed, stream, t, rows, cols):
    return torch.as_tensor(O.rng_u01(seed, stream, t, rows, cols).astype(np.float64))


def _dropout(x, rate, u):
    if rate <= 0.0:
        return x
    return x / (1.0 - rate) * (u >= np.float32(rate)).to(DT)


def _lstm(x, c_prev, h_prev, kernel, bias, fb):
    z = torch.cat([x, h_prev], dim=-1) @ kernel + bias
    i, j, f, o = torch.chunk(z, 4, dim=-1)
    c = torch.sigmoid(f + fb) * c_prev + torch.sigmoid(i) * torch.tanh(j)
    return c, torch.sigmoid(o) * torch.tanh(c)


def _zoneout_lstm(x, state, kernel, bias, zc, zh, fb, u_c, u_h):
    c_prev, h_prev = state
    new_c, new_h = _lstm(x, c_prev, h_prev, kernel, bias, fb)
    c = (1.0 - zc) * _dropout(new_c - c_prev, zc, u_c) + c_prev
    h = (1.0 - zh) * _dropout(new_h - h_prev, zh, u_h) + h_prev
    return new_h, (c, h)


STREAM_ENC_CONV = 20           # + layer index
STREAM_ZONE_ENC = 30           # fw: 30 (c), 31 (h); bw: 32, 33


def encoder_train(d, W, emb, lens, seed, drop_rate):
    """EncoderV2.call with is_training=True (tacotron_v2.py:53-63; Conv1d.call modules.py:149-157)."""
    x = emb
    B, T, _ = x.shape
    stats = {}
    for i in range(d.enc_conv_layers):
        p = f"encoder/conv1d_{i}"
        k = W[f"{p}/kernel"]
        pad_l = (k.shape[0] - 1) // 2
        xp = F.pad(x.transpose(1, 2), (pad_l, k.shape[0] - 1 - pad_l))
        y = F.conv1d(xp, k.permute(2, 1, 0)).transpose(1, 2)
        mean = y.mean(dim=(0, 1))
        var = y.var(dim=(0, 1), unbiased=False)                     # batch statistics over ALL positions, padding included
        stats[p] = (mean.detach(), var.detach())
        y = (y - mean) / torch.sqrt(var + d.bn_eps) * W[f"{p}/batch_normalization/gamma"] + W[f"{p}/batch_normalization/beta"]
        y = torch.relu(y)
        u = _u(seed, STREAM_ENC_CONV + i, 0, B * T, y.shape[2]).reshape(B, T, -1)
        x = _dropout(y, drop_rate, u)
    H = d.enc_units // 2
    lens_t = torch.as_tensor(np.asarray(lens))
    outs = []
    for di, name in enumerate(("fw", "bw")):
        kern, bias = W[f"encoder/bidirectional_rnn/{name}/lstm_cell/kernel"], W[f"encoder/bidirectional_rnn/{name}/lstm_cell/bias"]
        c = torch.zeros(B, H, dtype=DT)
        h = torch.zeros(B, H, dtype=DT)
        out = [None] * T
        for s in range(T):
            if di == 0:
                xt = x[:, s]
            else:                                                   # tf.reverse_sequence: position len-1-s
                idx = (lens_t - 1 - s).clamp(min=0)
                xt = x[torch.arange(B), idx]
            u_c, u_h = _u(seed, STREAM_ZONE_ENC + 2 * di, s, B, H), _u(seed, STREAM_ZONE_ENC + 2 * di + 1, s, B, H)
            o, (c2, h2) = _zoneout_lstm(xt, (c, h), kern, bias, d.zoneout_cell, d.zoneout_output, d.forget_bias, u_c, u_h)
            live = (s < lens_t)[:, None]
            c, h = torch.where(live, c2, c), torch.where(live, h2, h)
            out[s] = torch.where(live, o, torch.zeros_like(o))
        out = torch.stack(out, dim=1)
        if di == 1:                                                 # reverse back per row
            rev = torch.zeros_like(out)
            for b in range(B):
                n = int(lens_t[b])
                rev[b, :n] = out[b, :n].flip(0)
            out = rev
        outs.append(out)
    return torch.cat(outs, dim=-1), stats


STREAM_POST_CONV = 40          # + layer index


def postnet_train(d, W, mel, seed, drop_rate):
    """PostNetV2.call with is_training=True (tacotron_v2.py:181-204; Conv1d.call modules.py:149-157): conv -> batch-statistics
    BN -> tanh (none on the last convolution) -> dropout, then Dense(out_units) + residual on the raw input."""
    x = mel
    B, T, _ = x.shape
    stats = {}
    for i in range(1, d.postnet_layers + 1):
        p = f"postnet/conv1d_{i}"
        k = W[f"{p}/kernel"]
        pad_l = (k.shape[0] - 1) // 2
        xp = F.pad(x.transpose(1, 2), (pad_l, k.shape[0] - 1 - pad_l))
        y = F.conv1d(xp, k.permute(2, 1, 0)).transpose(1, 2)
        mean = y.mean(dim=(0, 1))
        var = y.var(dim=(0, 1), unbiased=False)
        stats[p] = (mean.detach(), var.detach())
        y = (y - mean) / torch.sqrt(var + d.bn_eps) * W[f"{p}/batch_normalization/gamma"] + W[f"{p}/batch_normalization/beta"]
        if i < d.postnet_layers:
            y = torch.tanh(y)
        u = _u(seed, STREAM_POST_CONV + i - 1, 0, B * T, y.shape[2]).reshape(B, T, -1)
        x = _dropout(y, drop_rate, u)
    return mel + x @ W["postnet/projection/kernel"] + W["postnet/projection/bias"], stats


def decoder_train(d, W, memory, lens, target, seed, prenet_drop_rate, return_state=False):
    """DecoderV1.call(is_training=True) with the V2 cells: TrainingHelper feeds targets[:, t-1, -M*nf:]."""
    B, T, E = memory.shape
    M, r, nf = d.num_mels, d.outputs_per_step, d.n_feed_frame
    S = target.shape[1] // r
    tgt = target.reshape(B, S, M * r)
    lens_t = torch.as_tensor(np.asarray(lens))
    mask = torch.arange(T)[None, :] < lens_t[:, None]
    values = memory * mask[:, :, None].to(DT)
    keys = values @ W["attention/memory_layer/kernel"]
    A, D = d.attention_lstm_units, d.decoder_lstm_units
    z = lambda n: torch.zeros(B, n, dtype=DT)
    s_att, s1, s2 = (z(A), z(A)), (z(D), z(D)), (z(D), z(D))
    ctx, att_state = z(E), z(T)
    conv_k = W["attention/location_features_convolution/kernel"]       # [K, 1, F]
    pad_l = (conv_k.shape[0] - 1) // 2
    mels, stops, aligns = [], [], []
    for t in range(S):
        x = torch.zeros(B, M * nf, dtype=DT) if t == 0 else tgt[:, t - 1, -M * nf:]
        for i in range(len(d.prenet_units)):
            x = torch.relu(x @ W[f"decoder/prenet_{i}/dense/kernel"] + W[f"decoder/prenet_{i}/dense/bias"])
            x = _dropout(x, prenet_drop_rate, _u(seed, O.STREAM_PRENET + i, t, B, x.shape[1]))
        h_att, s_att = _zoneout_lstm(torch.cat([x, ctx], -1), s_att, W["decoder/attention_lstm/kernel"], W["decoder/attention_lstm/bias"],
                                     d.zoneout_cell, d.zoneout_output, d.forget_bias,
                                     _u(seed, O.STREAM_ZONE_ATT, t, B, A), _u(seed, O.STREAM_ZONE_ATT + 1, t, B, A))
        q = h_att @ W["attention/query_layer/kernel"]
        f = F.conv1d(F.pad(att_state[:, None, :], (pad_l, conv_k.shape[0] - 1 - pad_l)), conv_k.permute(2, 1, 0)).transpose(1, 2) \
            + W["attention/location_features_convolution/bias"]
        loc = f @ W["attention/location_features_layer/kernel"]
        e = (W["attention/attention_variable"] * torch.tanh(keys + q[:, None, :] + loc + W["attention/attention_bias"])).sum(-1)
        e = e.masked_fill(~mask, float("-inf"))
        a = torch.softmax(e, dim=-1)
        att_state = att_state + a if d.cumulative_weights else a
        ctx = torch.bmm(a[:, None, :], values)[:, 0]
        y1, s1 = _zoneout_lstm(torch.cat([h_att, ctx], -1), s1, W["decoder/lstm_1/kernel"], W["decoder/lstm_1/bias"],
                               d.zoneout_cell, d.zoneout_output, d.forget_bias,
                               _u(seed, O.STREAM_ZONE_L1, t, B, D), _u(seed, O.STREAM_ZONE_L1 + 1, t, B, D))
        y2, s2 = _zoneout_lstm(y1, s2, W["decoder/lstm_2/kernel"], W["decoder/lstm_2/bias"],
                               d.zoneout_cell, d.zoneout_output, d.forget_bias,
                               _u(seed, O.STREAM_ZONE_L2, t, B, D), _u(seed, O.STREAM_ZONE_L2 + 1, t, B, D))
        mels.append(y2 @ W["decoder/out_projection/kernel"] + W["decoder/out_projection/bias"])
        stops.append(y2 @ W["decoder/stop_token_projection/kernel"] + W["decoder/stop_token_projection/bias"])
        aligns.append(a)
    mel = torch.stack(mels, 1).reshape(B, S * r, M)
    if return_state:    # the MultiRNNCell state tuple after the last step (tacotron_v2.py:155-161), each LSTM state as (c, h)
        state = dict(attention_c=s_att[0], attention_h=s_att[1], lstm1_c=s1[0], lstm1_h=s1[1], lstm2_c=s2[0], lstm2_h=s2[1],
                     attention=ctx, alignments=a, attention_state=att_state)
        return mel, torch.stack(stops, 1), torch.stack(aligns, 0), state
    return mel, torch.stack(stops, 1), torch.stack(aligns, 0)


def weighted_loss(l, w, count=None):
    """tf.losses.compute_weighted_loss, SUM_BY_NONZERO_WEIGHTS.  count: the denominator of a larger (data-parallel) batch
    this one is a shard of -- then the value is this shard's share of that batch's loss."""
    wb = w.expand_as(l)
    n = (wb != 0).sum() if count is None else count
    return (l * wb).sum() / n if float(n) > 0 else (l * wb).sum() * 0.0


def losses(mel, stop, labels, counts=None):
    """models.py:58-61, 136-148."""
    mel_loss = weighted_loss((mel - _t(labels["mel"])).abs(), _t(labels["spec_loss_mask"])[:, :, None], None if counts is None else counts[0])
    x, zt = stop[..., 0], _t(labels["done"])
    ce = torch.clamp(x, min=0) - x * zt + torch.log1p(torch.exp(-x.abs()))
    done_loss = weighted_loss(ce, _t(labels["binary_loss_mask"]), None if counts is None else counts[1])
    return mel_loss, done_loss


def train_forward_backward(d, w, ids, lens, labels, seed, prenet_drop_rate=0.5, encoder_drop_rate=0.5, decoder_only_memory=None,
                           loss_counts=None):
    """One training-mode forward + backward.  Returns dict(loss, mel_loss, done_loss, mel, stop, alignment [S,B,T], memory,
    grads{name: ndarray}, bn_stats).  decoder_only_memory: skip the encoder and treat this memory as a leaf (its gradient is
    returned as grads['memory'])."""
    W = {k: _t(v).clone().requires_grad_(True) for k, v in w.items() if not k.startswith("postnet/")}
    bn_stats = {}
    if decoder_only_memory is not None:
        memory = _t(decoder_only_memory).clone().requires_grad_(True)
    else:
        emb = W["embedding"][torch.as_tensor(np.asarray(ids), dtype=torch.long)]
        memory, bn_stats = encoder_train(d, W, emb, lens, seed, encoder_drop_rate)
    mel, stop, align = decoder_train(d, W, memory, lens, _t(labels["mel"]), seed, prenet_drop_rate)
    mel_loss, done_loss = losses(mel, stop, labels, loss_counts)      # loss_counts: data-parallel shard of a larger batch
    loss = mel_loss + done_loss
    loss.backward()
    grads = {k: (v.grad.numpy().astype(np.float32) if v.grad is not None else np.zeros(tuple(v.shape), np.float32)) for k, v in W.items()}
    if decoder_only_memory is not None:
        grads["memory"] = memory.grad.numpy().astype(np.float32)
    f32 = lambda t: t.detach().numpy().astype(np.float32)
    return {"loss": float(loss.detach()), "mel_loss": float(mel_loss.detach()), "done_loss": float(done_loss.detach()), "mel": f32(mel), "stop": f32(stop),
            "alignment": f32(align), "memory": f32(memory), "grads": grads,
            "bn_stats": {k: (f32(a), f32(b)) for k, (a, b) in bn_stats.items()}}
