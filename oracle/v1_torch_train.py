"""CPU ORACLE for the training step of the Tacotron-1 PostNet model -- test infrastructure, NOT product code.

A differentiable torch (float64, CPU) restatement of TacotronV1PostNetModel.model_fn(mode=TRAIN)'s forward
(tacotron/models.py:190-211): PostNet(is_training=True) = CBHG (tacotron/modules.py:163-232: conv bank with batch-statistics
batch norm, max-pool, two projections, residual, adjustment Dense, highway layers, bidirectional GRU over ALL positions -- the
PostNet calls the CBHG without lengths, tacotron_v1.py:181) -> Dense(num_freq) (tacotron_v1.py:176,182), and the masked L1
loss with SUM_BY_NONZERO_WEIGHTS (models.py:265-274).  Gradients come from torch.autograd: the checker of the hand-written HIP
backward pass (csrc/v1_train.h).  TF op semantics as SURVEY.md Appendix A (GRUCell A2: the reset gate multiplies h before the
candidate matmul; SAME padding of even kernels A3; batch norm A4).

PARITY STATUS: parity unpinned at the TensorFlow boundary (see oracle/t2_oracle.py); the inference-mode forward of the same
blocks is oracle/v1_oracle.py, and tests/test_oracle_v1.py checks this file's forward against it (batch statistics replaced by
the moving ones) and its gradients by finite differences."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=DT)


def conv1d_bn(x, W, p, act, eps, stats, use_moving=False):
    """Conv1d.call (modules.py:149-157): conv1d(SAME, no bias) -> batch norm (batch statistics over all B*T positions when
    training) -> activation.  drop_rate is 0 inside the CBHG (modules.py:171-198 pass none)."""
    k = W[f"{p}/kernel"]                                   # [K, Cin, Cout]
    pad_l = (k.shape[0] - 1) // 2
    xp = F.pad(x.transpose(1, 2), (pad_l, k.shape[0] - 1 - pad_l))
    y = F.conv1d(xp, k.permute(2, 1, 0)).transpose(1, 2)
    if use_moving:
        mean, var = W[f"{p}/batch_normalization/moving_mean"], W[f"{p}/batch_normalization/moving_variance"]
    else:
        mean = y.mean(dim=(0, 1))
        var = y.var(dim=(0, 1), unbiased=False)
        stats[p] = (mean.detach(), var.detach())
    y = (y - mean) / torch.sqrt(var + eps) * W[f"{p}/batch_normalization/gamma"] + W[f"{p}/batch_normalization/beta"]
    if act is torch.relu:
        return _relu(stats, y)
    return act(y) if act is not None else y


def _note_kink(stats, pre):
    """Smallest |pre-activation| fed to a ReLU so far: how close the loss is to a kink of its piecewise-smooth surface.  An fp32
    forward pass whose rounding lands an element on the other side of zero differentiates the neighbouring piece: both gradients
    are then right for their own side, so gradient comparisons are made at points whose margin is well above fp32 rounding."""
    stats["relu_margin"] = min(stats.get("relu_margin", float("inf")), float(pre.detach().abs().min()))


class _ReluShift(torch.autograd.Function):
    """relu whose DERIVATIVE is taken at z + shift: the gradient of the neighbouring smooth piece for every input within |shift|
    of the kink (the forward value is the plain relu's)."""

    @staticmethod
    def forward(ctx, z, shift):
        ctx.save_for_backward(z)
        ctx.shift = shift
        return z.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return g * (z + ctx.shift > 0).to(g.dtype), None


def _relu(stats, z):
    _note_kink(stats, z)
    shift = stats.get("relu_shift", 0.0)
    return torch.relu(z) if shift == 0.0 else _ReluShift.apply(z, shift)


def gru_seq(x, W, p, reverse, lens=None):
    """dynamic_rnn over tf.nn.rnn_cell.GRUCell (App. A2, A8).  lens: per-row lengths -- a row's steps past its length copy the
    state through and emit zeros; the backward direction of bidirectional_dynamic_rnn walks a row from its last live position
    (reverse_sequence with the lengths) and its outputs land at the positions they belong to."""
    B, T, _ = x.shape
    wg, bg, wc, bc = W[f"{p}/gates/kernel"], W[f"{p}/gates/bias"], W[f"{p}/candidate/kernel"], W[f"{p}/candidate/bias"]
    H = wc.shape[1]
    lens_t = torch.full((B,), T, dtype=torch.long) if lens is None else torch.as_tensor(np.asarray(lens), dtype=torch.long)
    rows = torch.arange(B)
    h = torch.zeros(B, H, dtype=DT)
    out = torch.zeros(B, T, H, dtype=DT)
    for s in range(T):
        live = s < lens_t                                                  # step s of every row
        pos = (lens_t - 1 - s).clamp(min=0) if reverse else torch.full((B,), s, dtype=torch.long)
        xt = x[rows, pos]
        g = torch.sigmoid(torch.cat([xt, h], -1) @ wg + bg)
        r, u = g[:, :H], g[:, H:]
        c = torch.tanh(torch.cat([xt, r * h], -1) @ wc + bc)
        hn = u * h + (1.0 - u) * c
        h = torch.where(live[:, None], hn, h)
        out = out.index_put((rows[live], pos[live]), hn[live])
    return out


def cbhg_train(W, p, x, max_filter_width, num_highway, eps, stats, use_moving=False, lens=None):
    """CBHG.call (modules.py:205-229), is_training=True, input_lengths=None."""
    bank = torch.cat([conv1d_bn(x, W, f"{p}/conv1d_K{k}", torch.relu, eps, stats, use_moving) for k in range(1, max_filter_width + 1)], dim=-1)
    pooled = torch.cat([torch.maximum(bank[:, :-1], bank[:, 1:]), bank[:, -1:]], dim=1)        # MaxPooling1D(2, 1, SAME), App. A3
    p1 = conv1d_bn(pooled, W, f"{p}/proj1", torch.relu, eps, stats, use_moving)
    p2 = conv1d_bn(p1, W, f"{p}/proj2", None, eps, stats, use_moving)
    hw = p2 + x
    if f"{p}/adjustment/kernel" in W:
        hw = hw @ W[f"{p}/adjustment/kernel"] + W[f"{p}/adjustment/bias"]
    for i in range(1, num_highway + 1):
        hp = f"{p}/highway_{i}"
        zh = hw @ W[f"{hp}/H/kernel"] + W[f"{hp}/H/bias"]
        h = _relu(stats, zh)
        t = torch.sigmoid(hw @ W[f"{hp}/T/kernel"] + W[f"{hp}/T/bias"])
        hw = h * t + hw * (1.0 - t)
    rp = f"{p}/bidirectional_rnn"
    return torch.cat([gru_seq(hw, W, f"{rp}/fw/gru_cell", False, lens), gru_seq(hw, W, f"{rp}/bw/gru_cell", True, lens)], dim=-1)


def postnet_train_forward_backward(d, w, mel, spec, spec_loss_mask, loss_count=None, use_moving=False, relu_shift=0.0):
    """One training-mode forward + backward of the PostNet model.  Returns dict(loss, spec [B, T, num_freq], grads{name},
    bn_stats{prefix: (mean, biased variance)}, relu_margin = smallest |ReLU input| of the pass).  relu_shift != 0: the gradient with every ReLU input within |relu_shift| of zero
    differentiated on the other side (``_ReluShift``) -- the spread between shifts -s, 0, +s is how far the gradient can move when
    fp32 rounding of the forward pass crosses such kinks.  loss_count: the denominator of a larger (data-parallel) batch."""
    pre = "v1/postnet/"
    W = {k: _t(v).clone().requires_grad_(not k.endswith(("moving_mean", "moving_variance"))) for k, v in w.items() if k.startswith(pre)}
    stats = {"relu_shift": float(relu_shift)} if relu_shift else {}
    y = cbhg_train(W, pre + "cbhg", _t(mel), d.post_net_max_filter_width, d.post_net_num_highway, d.bn_eps, stats, use_moving)
    out = y @ W[pre + "linear/kernel"] + W[pre + "linear/bias"]
    l1 = (out - _t(spec)).abs()
    wb = _t(spec_loss_mask)[:, :, None].expand_as(l1)
    n = (wb != 0).sum() if loss_count is None else loss_count
    loss = (l1 * wb).sum() / n
    loss.backward()
    grads = {k: (v.grad.numpy().astype(np.float32) if v.grad is not None else np.zeros(tuple(v.shape), np.float32)) for k, v in W.items()}
    f32 = lambda t: t.detach().numpy().astype(np.float32)
    margin = stats.pop("relu_margin", float("inf"))
    stats.pop("relu_shift", None)
    return {"loss": float(loss.detach()), "spec": f32(out), "grads": grads, "bn_stats": {k: (f32(a), f32(b)) for k, (a, b) in stats.items()},
            "relu_margin": margin}


def encoder_train_forward_backward(d, w, ids, lens, dmemory, drop_rate=0.5, seed=0, relu_shift=0.0):
    """Embedding + EncoderV1(is_training=True) (models.py:40-47, tacotron_v1.py:18-49): two PreNets (Dense + relu + dropout,
    modules.py:80-91; masks from the shared counter-based generator, stream 50 + layer, step 0, row = position in the batch) and
    the CBHG with batch-statistics batch norm.  Scalar for the gradients: sum(memory * dmemory), i.e. dmemory is d loss / d memory.
    Returns dict(memory, grads{name}, bn_stats)."""
    from oracle import t2_oracle as O
    names = [k for k in w if k == "embedding" or k.startswith("v1/encoder/")]
    W = {k: _t(w[k]).clone().requires_grad_(not k.endswith(("moving_mean", "moving_variance"))) for k in names}
    ids = np.asarray(ids)
    B, T = ids.shape
    x = W["embedding"][torch.as_tensor(ids.reshape(-1), dtype=torch.long)]
    keep = 1.0 - drop_rate
    for i in range(2):
        q = f"v1/encoder/prenet_{i}/dense"
        x = torch.relu(x @ W[q + "/kernel"] + W[q + "/bias"])
        if drop_rate > 0:
            u = O.rng_u01(seed, 50 + i, 0, B * T, x.shape[1])
            x = x / keep * _t((u >= np.float32(drop_rate)).astype(np.float64))
    stats = {"relu_shift": float(relu_shift)} if relu_shift else {}
    y = cbhg_train(W, "v1/encoder/cbhg", x.reshape(B, T, -1), d.max_filter_width, d.num_highway, d.bn_eps, stats, False, lens)
    (y * _t(dmemory)).sum().backward()
    stats.pop("relu_margin", None); stats.pop("relu_shift", None)
    f32 = lambda t: t.detach().numpy().astype(np.float32)
    grads = {k: (f32(v.grad) if v.grad is not None else np.zeros(tuple(v.shape), np.float32)) for k, v in W.items()}
    return {"memory": f32(y), "grads": grads, "bn_stats": {k: (f32(a), f32(b)) for k, (a, b) in stats.items()}}


# ---- DecoderV1 training step (tacotron_v1.py:56-153 with is_training=True; losses of models.py:58-61) ----
def _gru_cell(x, h, W, p):
    wg, bg, wc, bc = W[f"{p}/gates/kernel"], W[f"{p}/gates/bias"], W[f"{p}/candidate/kernel"], W[f"{p}/candidate/bias"]
    H = wc.shape[1]
    g = torch.sigmoid(torch.cat([x, h], -1) @ wg + bg)
    r, u = g[:, :H], g[:, H:]
    c = torch.tanh(torch.cat([x, r * h], -1) @ wc + bc)
    return u * h + (1.0 - u) * c


def decoder_train_forward_backward(d, w, memory, lens, labels, drop_rate=0.5, seed=0, loss_counts=None):
    """DecoderV1.call(is_training=True) (TrainingHelper: the last n_feed_frame target frames of the previous group are fed,
    helpers.py:141-184) on OutputAndStopTokenWrapper(DecoderRNNV1(OutputProjectionWrapper(AttentionRNN(GRUCell, PreNets,
    BahdanauAttention)), 2 x ResidualWrapper(GRUCell))), and loss = spec_loss(mel) + binary_loss(stop) (models.py:58-61).
    PreNet dropout masks from the shared counter-based generator: stream 60 + layer, step t, row, unit.
    labels: mel [B, S*r, M], done [B, S], spec_loss_mask [B, S*r], binary_loss_mask [B, S].
    Returns dict(mel, stop, alignments [S, B, T], loss, mel_loss, done_loss, grads{name}, dmemory)."""
    from oracle import t2_oracle as O
    names = [k for k in w if k.startswith(("v1/decoder/", "v1/attention/"))]
    W = {k: _t(w[k]).clone().requires_grad_(True) for k in names}
    mem = _t(memory).clone().requires_grad_(True)
    B, T, E = mem.shape
    M, r, nf = d.num_mels, d.outputs_per_step, d.n_feed_frame
    target = _t(labels["mel"])
    S = target.shape[1] // r
    grouped = target.reshape(B, S, M * r)
    lens_t = torch.full((B,), T, dtype=torch.long) if lens is None else torch.as_tensor(np.asarray(lens), dtype=torch.long)
    live = (torch.arange(T)[None, :] < lens_t[:, None])
    values = mem * live[:, :, None].to(DT)
    keys = values @ W["v1/attention/memory_layer/kernel"]
    A, D = d.attention_out_units, d.decoder_out_units
    ha, ctx, h1, h2 = torch.zeros(B, A, dtype=DT), torch.zeros(B, E, dtype=DT), torch.zeros(B, D, dtype=DT), torch.zeros(B, D, dtype=DT)
    keep = 1.0 - drop_rate
    mels, stops, aligns = [], [], []
    for t in range(S):
        x = torch.zeros(B, M * nf, dtype=DT) if t == 0 else grouped[:, t - 1, -M * nf:]
        for i in range(2):
            q = f"v1/decoder/prenet_{i}/dense"
            x = torch.relu(x @ W[q + "/kernel"] + W[q + "/bias"])
            if drop_rate > 0:
                u = O.rng_u01(seed, 60 + i, t, B, x.shape[1])
                x = x / keep * _t((u >= np.float32(drop_rate)).astype(np.float64))
        ha = _gru_cell(torch.cat([x, ctx], -1), ha, W, "v1/decoder/attention_gru")
        qv = ha @ W["v1/attention/query_layer/kernel"]
        score = (W["v1/attention/attention_v"] * torch.tanh(keys + qv[:, None, :])).sum(-1)
        score = torch.where(live, score, torch.full_like(score, -float("inf")))
        a = torch.softmax(score, dim=-1)
        ctx = (a[:, :, None] * values).sum(1)
        y0 = torch.cat([ha, ctx], -1) @ W["v1/decoder/output_projection/kernel"] + W["v1/decoder/output_projection/bias"]
        h1 = _gru_cell(y0, h1, W, "v1/decoder/gru_1")
        y1 = y0 + h1
        h2 = _gru_cell(y1, h2, W, "v1/decoder/gru_2")
        y2 = y1 + h2
        mels.append(y2 @ W["v1/decoder/out_projection/kernel"] + W["v1/decoder/out_projection/bias"])
        stops.append(y2 @ W["v1/decoder/stop_token_projection/kernel"] + W["v1/decoder/stop_token_projection/bias"])
        aligns.append(a)
    mel = torch.stack(mels, 1).reshape(B, S * r, M)
    stop = torch.stack(stops, 1)[:, :, 0]
    m1, m2, done = _t(labels["spec_loss_mask"]), _t(labels["binary_loss_mask"]), _t(labels["done"])
    n1 = float((m1 != 0).sum() * M) if loss_counts is None else loss_counts[0]
    n2 = float((m2 != 0).sum()) if loss_counts is None else loss_counts[1]
    mel_loss = ((mel - target).abs() * m1[:, :, None]).sum() / n1
    ce = torch.clamp(stop, min=0) - stop * done + torch.log1p(torch.exp(-stop.abs()))
    done_loss = (ce * m2).sum() / n2
    loss = mel_loss + done_loss
    loss.backward()
    f32 = lambda t_: t_.detach().numpy().astype(np.float32)
    return {"mel": f32(mel), "stop": f32(stop), "alignments": f32(torch.stack(aligns, 0)), "loss": float(loss.detach()),
            "mel_loss": float(mel_loss.detach()), "done_loss": float(done_loss.detach()),
            "grads": {k: f32(v.grad) for k, v in W.items()}, "dmemory": f32(mem.grad)}
