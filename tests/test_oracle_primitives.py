"""Oracle primitives (NumPy restatement) against the independently written torch-CPU versions
(SURVEY.md §8c fixture kind 1) and the TF-semantics quirks of SURVEY.md Appendix A/B."""
import numpy as np
import pytest
import torch

from oracle import t2_oracle as O
from oracle import torch_check as TC
from tacotron2_amd.hparams import PROFILE_TINY, T2Dims
from tacotron2_amd.weights import init_weights

TOL = 2e-6


def rng(seed=0):
    return np.random.default_rng(seed)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 31])
def test_conv1d_same_odd_and_even(k):
    r = rng(k)
    x = r.uniform(-1, 1, (2, 9, 3)).astype(np.float32)
    w = r.uniform(-1, 1, (k, 3, 4)).astype(np.float32)
    b = r.uniform(-1, 1, (4,)).astype(np.float32)
    np.testing.assert_allclose(O.conv1d_same(x, w, b), TC.conv1d_same(x, w, b), atol=1e-5)
    # hand check of the SAME padding rule: pad_left = (k-1)//2
    pl = (k - 1) // 2
    t = 0
    ref = sum(w[j].T @ x[0, t + j - pl] for j in range(k) if 0 <= t + j - pl < 9) + b
    np.testing.assert_allclose(O.conv1d_same(x, w, b)[0, 0], ref, atol=1e-5)


def test_batch_norm_inference():
    r = rng(1)
    x = r.normal(size=(2, 7, 5)).astype(np.float32)
    g, b, m = (r.normal(size=5).astype(np.float32) for _ in range(3))
    v = r.uniform(0.5, 2, 5).astype(np.float32)
    np.testing.assert_allclose(O.batch_norm_inference(x, g, b, m, v), TC.batch_norm_inference(x, g, b, m, v), atol=TOL * 4)
    # fresh statistics: x / sqrt(1.001)  
    y = O.batch_norm_inference(x, np.ones(5, np.float32), np.zeros(5, np.float32), np.zeros(5, np.float32), np.ones(5, np.float32))
    np.testing.assert_allclose(y, x / np.sqrt(1.001), atol=TOL)


def test_lstm_cell_vs_torch():
    r = rng(2)
    B, I, H = 3, 5, 4
    x, c, h = (r.normal(size=s).astype(np.float32) for s in ((B, I), (B, H), (B, H)))
    k = r.normal(size=(I + H, 4 * H)).astype(np.float32) * 0.5
    b = r.normal(size=(4 * H,)).astype(np.float32)
    c1, h1 = O.lstm_cell(x, c, h, k, b)
    c2, h2 = TC.lstm_cell(x, c, h, k, b)
    np.testing.assert_allclose(c1, c2, atol=TOL)
    np.testing.assert_allclose(h1, h2, atol=TOL)


def test_zoneout_returns_raw_output_but_zoned_state():
    """modules.py:263,283."""
    r = rng(3)
    B, I, H = 2, 3, 4
    x, c, h = (r.normal(size=s).astype(np.float32) for s in ((B, I), (B, H), (B, H)))
    k = r.normal(size=(I + H, 4 * H)).astype(np.float32)
    b = np.zeros(4 * H, np.float32)
    new_c, new_h = O.lstm_cell(x, c, h, k, b)
    out, (zc, zh) = O.zoneout_lstm_cell(x, (c, h), k, b, 0.25, 0.5)
    np.testing.assert_array_equal(out, new_h)
    np.testing.assert_allclose(zc, 0.75 * new_c + 0.25 * c, atol=TOL)
    np.testing.assert_allclose(zh, 0.5 * new_h + 0.5 * h, atol=TOL)


def test_bidirectional_rnn_vs_torch_packed():
    """A8: zero outputs / frozen state beyond the length, bw on reverse_sequence."""
    d = PROFILE_TINY.replace(zoneout_cell=0.0, zoneout_output=0.0, enc_conv_layers=0, embedding_dim=10)
    w = init_weights(d, 5, perturb=True)
    r = rng(4)
    x = r.normal(size=(3, 7, 10)).astype(np.float32)
    lens = np.array([7, 4, 1])
    got = O.encoder_v2(d, w, x, lens)
    p = "encoder/bidirectional_rnn"
    ref = TC.bidirectional_lstm(x, lens, w[f"{p}/fw/lstm_cell/kernel"], w[f"{p}/fw/lstm_cell/bias"],
                                w[f"{p}/bw/lstm_cell/kernel"], w[f"{p}/bw/lstm_cell/bias"])
    np.testing.assert_allclose(got, ref, atol=1e-5)
    assert np.all(got[1, 4:] == 0) and np.all(got[2, 1:] == 0)


def test_lsa_step_vs_torch_and_quirks():
    d = PROFILE_TINY
    w = init_weights(d, 7, perturb=True)
    r = rng(5)
    B, T, E = 3, 9, d.enc_units
    memory = r.normal(size=(B, T, E)).astype(np.float32)
    lens = np.array([9, 6, 2])
    mech = O.LocationSensitiveAttention(d, w, memory, lens)
    # values are the length-masked memory, keys = values . W_mem (App. A5)
    assert np.all(mech.values[1, 6:] == 0) and np.all(mech.keys[2, 2:] == 0)
    query = r.normal(size=(B, d.attention_lstm_units)).astype(np.float32)
    state = mech.initial_state(B)
    for _ in range(3):
        al, nxt = mech(query, state)
        al2, nxt2, _ = TC.lsa_step(query, state, mech.keys, mech.values, lens,
                                   w["attention/query_layer/kernel"],
                                   w["attention/location_features_convolution/kernel"],
                                   w["attention/location_features_convolution/bias"],
                                   w["attention/location_features_layer/kernel"],
                                   w["attention/attention_variable"], w["attention/attention_bias"])
        np.testing.assert_allclose(al, al2, atol=TOL)
        np.testing.assert_allclose(nxt, nxt2, atol=TOL)
        # masked positions get exactly zero probability; rows sum to one
        assert np.all(al[1, 6:] == 0) and np.all(al[2, 2:] == 0)
        np.testing.assert_allclose(al.sum(-1), 1.0, atol=1e-6)
        # cumulative state (tacotron_v2.py:137-140)
        np.testing.assert_allclose(nxt, al + state, atol=TOL)
        state = nxt
    with pytest.raises(TypeError):
        O.LocationSensitiveAttention(d, w, memory, lens, smoothing=True)


def test_embedding_range_assert():
    table = np.zeros((4, 2), np.float32)
    with pytest.raises(AssertionError):
        O.embedding_lookup(np.array([[0, 4]]), table)
    with pytest.raises(AssertionError):
        O.embedding_lookup(np.array([[-1]]), table)


def test_losses_and_lr():
    y = np.zeros((2, 4, 3), np.float32)
    yh = np.ones((2, 4, 3), np.float32)
    mask = np.array([[1, 1, 0, 0], [1, 1, 1, 0]], np.float32)
    assert abs(O.spec_loss(yh, y, mask) - 1.0) < 1e-6            # sum / (5 * 3 nonzero weights)
    x = np.array([[[2.0], [-1.0]]], np.float32)
    z = np.array([[1.0, 0.0]], np.float32)
    m = np.array([[1.0, 1.0]], np.float32)
    ref = np.mean([np.log1p(np.exp(-2.0)), np.log1p(np.exp(-1.0))])
    assert abs(O.binary_loss(x, z, m) - ref) < 1e-6
    assert abs(O.learning_rate_decay(0.002, 3999) - 0.002) < 1e-9
    assert abs(O.learning_rate_decay(0.002, 0) - 0.002 * 4000 ** 0.5 * 4000 ** -1.5) < 1e-12


def test_dims_validation():
    with pytest.raises(ValueError):
        T2Dims(zoneout_cell=1.5)
    with pytest.raises(AssertionError):
        T2Dims(enc_units=511)
    with pytest.raises(AssertionError):
        T2Dims(n_feed_frame=3, outputs_per_step=2)


def test_optimizer_block_oracle_against_torch():
    """clip_by_global_norm and TF-form Adam (epsilon outside the sqrt) against an independent torch
    statement of the same formulas."""
    rng = np.random.default_rng(3)
    n = 257
    p, g = rng.standard_normal(n).astype(np.float32), (rng.standard_normal(n) * 0.5).astype(np.float32)
    (gc,), norm = O.clip_by_global_norm([g], 1.0)
    tg = torch.tensor(g, dtype=torch.float64)
    assert abs(norm - float(tg.norm())) < 1e-9
    np.testing.assert_allclose(gc, (tg * (1.0 / max(float(tg.norm()), 1.0))).numpy(), rtol=1e-6)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    tp, tm, tv = torch.tensor(p, dtype=torch.float64), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    for t in range(1, 4):
        p, m, v = O.adam_apply(p, gc, m, v, t, 0.01)
        tgc = torch.tensor(gc, dtype=torch.float64)
        tm = 0.9 * tm + 0.1 * tgc
        tv = 0.999 * tv + 0.001 * tgc * tgc
        tp = tp - 0.01 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * tm / (tv.sqrt() + 1e-8)
        np.testing.assert_allclose(p, tp.numpy(), atol=2e-6)
    # a small gradient is left alone by the clipping
    (gs,), _ = O.clip_by_global_norm([g * 1e-3], 1.0)
    np.testing.assert_array_equal(gs, (g * 1e-3).astype(np.float32))
