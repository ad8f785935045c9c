"""CPU ORACLE support -- test infrastructure, NOT product code.

An independently written torch-CPU implementation of the primitives in ``t2_oracle.py`` using
torch's own library ops (F.conv1d, F.batch_norm, torch LSTMCell / packed bidirectional nn.LSTM
after the TF->torch gate permutation, torch.softmax).  Two independently written
implementations agreeing to ~1e-6 is the strongest parity anchor available here, because
TensorFlow cannot run anywhere in this pipeline.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).float()


def conv1d_same(x, kernel, bias=None):
    """TF SAME padding: left (k-1)//2, right the rest; cross-correlation; kernel [k,in,out]."""
    k = kernel.shape[0]
    pl = (k - 1) // 2
    xt = F.pad(t(x).transpose(1, 2), (pl, k - 1 - pl))
    wt = t(kernel).permute(2, 1, 0).contiguous()           # [out, in, k]
    y = F.conv1d(xt, wt, None if bias is None else t(bias))
    return y.transpose(1, 2).contiguous().numpy()


def batch_norm_inference(x, gamma, beta, mean, var, eps=1e-3):
    y = F.batch_norm(t(x).transpose(1, 2), t(mean), t(var), t(gamma), t(beta), training=False, eps=eps)
    return y.transpose(1, 2).contiguous().numpy()


def _tf_lstm_to_torch(kernel, bias, n_in, forget_bias=1.0):
    """TF kernel [in+h, 4h] with column blocks i|j|f|o -> torch weight_ih [4h,in], weight_hh [4h,h]
    with row blocks i|f|g|o; forget_bias folded into the f bias."""
    H = kernel.shape[1] // 4
    i, j, f, o = np.split(kernel, 4, axis=1)
    bi, bj, bf, bo = np.split(bias, 4)
    wk = np.concatenate([i, f, j, o], axis=1)
    b = np.concatenate([bi, bf + forget_bias, bj, bo])
    return t(wk[:n_in].T), t(wk[n_in:].T), t(b), H


def lstm_cell(x, c_prev, h_prev, kernel, bias, forget_bias=1.0):
    w_ih, w_hh, b, H = _tf_lstm_to_torch(kernel, bias, x.shape[1], forget_bias)
    h, c = torch._VF.lstm_cell(t(x), (t(h_prev), t(c_prev)), w_ih, w_hh, b, torch.zeros_like(b))
    return c.numpy(), h.numpy()


def bidirectional_lstm(x, lengths, kfw, bfw, kbw, bbw, forget_bias=1.0):
    """Packed bidirectional nn.LSTM == tf.nn.bidirectional_dynamic_rnn(sequence_length) for plain
    (zoneout-free) LSTM cells; padded outputs are zero."""
    B, T, cin = x.shape
    H = kfw.shape[1] // 4
    lstm = torch.nn.LSTM(cin, H, batch_first=True, bidirectional=True)
    with torch.no_grad():
        for sfx, (k, b) in (("", (kfw, bfw)), ("_reverse", (kbw, bbw))):
            w_ih, w_hh, bb, _ = _tf_lstm_to_torch(k, b, cin, forget_bias)
            getattr(lstm, "weight_ih_l0" + sfx).copy_(w_ih)
            getattr(lstm, "weight_hh_l0" + sfx).copy_(w_hh)
            getattr(lstm, "bias_ih_l0" + sfx).copy_(bb)
            getattr(lstm, "bias_hh_l0" + sfx).zero_()
        packed = torch.nn.utils.rnn.pack_padded_sequence(t(x), torch.as_tensor(np.asarray(lengths)),
                                                         batch_first=True, enforce_sorted=False)
        out, _ = lstm(packed)
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    return out.numpy()


def lsa_step(query, prev_state, keys, values, lengths, wq, conv_k, conv_b, wloc, v_a, b_a, cumulative=True):
    """One location-sensitive attention step written with torch ops only."""
    q = t(query) @ t(wq)
    K = conv_k.shape[0]
    pl = (K - 1) // 2
    a = F.pad(t(prev_state)[:, None, :], (pl, K - 1 - pl))
    f = F.conv1d(a, t(conv_k).permute(2, 1, 0).contiguous(), t(conv_b)).transpose(1, 2)  # [B,T,F]
    pf = f @ t(wloc)
    e = (t(v_a) * torch.tanh(t(keys) + q[:, None, :] + pf + t(b_a))).sum(-1)
    if lengths is not None:
        mask = torch.arange(e.shape[1])[None, :] < torch.as_tensor(np.asarray(lengths))[:, None]
        e = e.masked_fill(~mask, float("-inf"))
    al = torch.softmax(e, dim=-1)
    ctx = torch.bmm(al[:, None, :], t(values))[:, 0]
    nxt = al + t(prev_state) if cumulative else al
    return al.numpy(), nxt.numpy(), ctx.numpy()


# ---- Tacotron v1 primitives (independent statements for oracle/v1_oracle.py) -----------------------
def gru_cell_tf(x, h, wg, bg, wc, bc):
    """TF GRUCell written from the equations of SURVEY.md App. A2 with torch ops (NOT torch.nn.GRU, whose
    candidate applies the reset gate after the matmul)."""
    x, h = t(x), t(h)
    n_in = x.shape[1]
    wg, wc = t(wg), t(wc)
    gates = torch.sigmoid(x @ wg[:n_in] + h @ wg[n_in:] + t(bg))
    H = h.shape[1]
    r, u = gates[:, :H], gates[:, H:]
    cand = torch.tanh(x @ wc[:n_in] + (r * h) @ wc[n_in:] + t(bc))
    return (u * h + (1 - u) * cand).numpy()


def maxpool2_same(x):
    """MaxPooling1D(2, strides=1, 'SAME') through torch.nn.functional.max_pool1d with -inf right padding."""
    xt = t(x).transpose(1, 2)
    xt = F.pad(xt, (0, 1), value=float("-inf"))
    return F.max_pool1d(xt, kernel_size=2, stride=1).transpose(1, 2).numpy()


def highway(x, wh, bh, wt, bt):
    x = t(x)
    h = torch.relu(F.linear(x, t(wh).T, t(bh)))
    g = torch.sigmoid(F.linear(x, t(wt).T, t(bt)))
    return (h * g + x * (1 - g)).numpy()


def bahdanau_step(query, keys, values, lengths, wq, v):
    q = t(query) @ t(wq)
    score = (t(v) * torch.tanh(t(keys) + q[:, None, :])).sum(-1)
    if lengths is not None:
        T = keys.shape[1]
        mask = torch.arange(T)[None, :] < torch.as_tensor(np.asarray(lengths))[:, None]
        score = score.masked_fill(~mask, float("-inf"))
    a = torch.softmax(score, dim=-1)
    ctx = torch.bmm(a[:, None, :], t(values))[:, 0]
    return a.numpy(), ctx.numpy()
