"""The training-step oracle (oracle/torch_train.py): forward against the NumPy oracle's training-mode decoder with the
same sampled masks, gradients against central finite differences.  CPU only."""
import numpy as np
import pytest

from oracle import t2_oracle as O
from oracle import torch_train as TT
from tacotron2_amd.hparams import PROFILE_TINY
from tacotron2_amd.weights import init_weights


def _case(seed=0):
    d = PROFILE_TINY.replace(outputs_per_step=2, n_feed_frame=2)
    w = init_weights(d, 3, perturb=True)
    rng = np.random.default_rng(seed)
    B, T, S = 3, 6, 4
    ids = rng.integers(0, d.num_symbols, (B, T))
    lens = np.array([6, 4, 5])
    tgt = rng.uniform(0, 1, (B, S * 2, d.num_mels)).astype(np.float32)
    tl = np.array([8, 6, 4])
    labels = {"mel": tgt, "done": (np.arange(S)[None, :] >= (tl[:, None] // 2 - 1)).astype(np.float32),
              "spec_loss_mask": (np.arange(S * 2)[None, :] < tl[:, None]).astype(np.float32),
              "binary_loss_mask": (np.arange(S)[None, :] < (tl[:, None] // 2)).astype(np.float32)}
    return d, w, ids, lens, labels


def test_training_forward_matches_numpy_oracle_with_the_same_masks():
    d, w, ids, lens, labels = _case()
    memory = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    out = TT.train_forward_backward(d, w, ids, lens, labels, seed=9, prenet_drop_rate=0.5, decoder_only_memory=memory)
    mel, stop, st = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=labels["mel"], train_seed=9,
                                 prenet_drop_rate=0.5)
    np.testing.assert_allclose(out["mel"], mel, atol=2e-5)
    np.testing.assert_allclose(out["stop"], stop, atol=2e-5)
    np.testing.assert_allclose(out["alignment"], np.stack(st[0].alignment_history), atol=2e-5)
    assert abs(out["mel_loss"] - float(O.spec_loss(mel, labels["mel"], labels["spec_loss_mask"]))) < 1e-5
    assert abs(out["done_loss"] - float(O.binary_loss(stop, labels["done"], labels["binary_loss_mask"]))) < 1e-5
    # the sampled path differs from the deterministic one
    mel_i, _, _ = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=labels["mel"])
    assert np.abs(mel_i - mel).max() > 1e-3


def test_gradients_against_finite_differences():
    d, w, ids, lens, labels = _case(1)
    base = TT.train_forward_backward(d, w, ids, lens, labels, seed=4, prenet_drop_rate=0.5, encoder_drop_rate=0.25)
    rng = np.random.default_rng(2)
    eps = 1e-4
    for name in ("decoder/lstm_1/kernel", "attention/location_features_convolution/kernel", "attention/attention_variable",
                 "decoder/prenet_0/dense/kernel", "encoder/conv1d_0/kernel", "encoder/bidirectional_rnn/bw/lstm_cell/kernel",
                 "encoder/conv1d_1/batch_normalization/gamma", "embedding", "decoder/stop_token_projection/bias"):
        g = base["grads"][name]
        for _ in range(2):
            idx = tuple(int(rng.integers(0, n)) for n in g.shape)
            vals = []
            for sgn in (+1, -1):
                w2 = dict(w)
                a = w[name].astype(np.float64).copy()
                a[idx] += sgn * eps
                w2[name] = a
                vals.append(TT.train_forward_backward(d, w2, ids, lens, labels, seed=4, prenet_drop_rate=0.5, encoder_drop_rate=0.25)["loss"])
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - g[idx]) < 1e-5 + 1e-3 * abs(fd), (name, idx, fd, g[idx])
