"""The V1 oracle (oracle/v1_oracle.py) against independently written torch-CPU primitives and the reference's own
property test (tests/decoder_test.py:42-67).  CPU only."""
import numpy as np
import pytest

from oracle import t2_oracle as O
from oracle import torch_check as TC
from oracle import v1_oracle as V
from tacotron2_amd.hparams import V1_PROFILE, V1_TINY
from tacotron2_amd.weights_v1 import init_weights_v1, weight_shapes_v1


def test_gru_cell_is_the_tf_variant():
    rng = np.random.default_rng(0)
    B, n_in, H = 4, 5, 6
    x, h = rng.standard_normal((B, n_in)).astype(np.float32), rng.standard_normal((B, H)).astype(np.float32)
    wg, bg = rng.standard_normal((n_in + H, 2 * H)).astype(np.float32) * 0.4, rng.standard_normal(2 * H).astype(np.float32)
    wc, bc = rng.standard_normal((n_in + H, H)).astype(np.float32) * 0.4, rng.standard_normal(H).astype(np.float32)
    got = V.gru_cell(x, h, wg, bg, wc, bc)
    np.testing.assert_allclose(got, TC.gru_cell_tf(x, h, wg, bg, wc, bc), atol=1e-6)
    # hand-computed known answer: one unit, zero input -> h' = u*h + (1-u)*tanh(wc_h * r*h)
    one = lambda v: np.array([[v]], np.float32)
    r = u = 1 / (1 + np.exp(-(0.5 * 2.0 + 1.0)))
    want = u * 2.0 + (1 - u) * np.tanh(0.25 * r * 2.0)
    got = V.gru_cell(np.zeros((1, 1), np.float32), one(2.0), np.array([[0, 0], [0.5, 0.5]], np.float32), np.ones(2, np.float32),
                     np.array([[0], [0.25]], np.float32), np.zeros(1, np.float32))
    assert abs(float(got[0, 0]) - want) < 1e-6


def test_maxpool_highway_bahdanau():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 6, 4)).astype(np.float32)
    np.testing.assert_array_equal(V.maxpool2_same(x), TC.maxpool2_same(x))
    assert np.array_equal(V.maxpool2_same(x)[:, -1], x[:, -1])                # last frame passes through
    w = {f"p/{g}/{k}": rng.standard_normal(s).astype(np.float32) for g in "HT" for k, s in (("kernel", (4, 4)), ("bias", (4,)))}
    np.testing.assert_allclose(V.highway(x, w, "p"), TC.highway(x, w["p/H/kernel"], w["p/H/bias"], w["p/T/kernel"], w["p/T/bias"]), atol=1e-6)
    d = V1_TINY
    wt = init_weights_v1(d, 3, perturb=True)
    mem = rng.standard_normal((3, 7, d.cbhg_out_units)).astype(np.float32)
    lens = np.array([7, 3, 5])
    mech = V.BahdanauAttention(wt, mem, lens)
    q = rng.standard_normal((3, d.attention_out_units)).astype(np.float32)
    a, _ = mech(q, mech.initial_state(3))
    a2, ctx2 = TC.bahdanau_step(q, mech.keys, mech.values, lens, wt["v1/attention/query_layer/kernel"], wt["v1/attention/attention_v"])
    np.testing.assert_allclose(a, a2, atol=1e-6)
    assert np.all(a[1, 3:] == 0) and abs(a[1].sum() - 1) < 1e-6


def test_even_kernel_padding_of_the_bank():
    """The CBHG bank has even kernels: SAME pads (k-1)//2 on the left, the remainder on the right (App. A3)."""
    x = np.zeros((1, 6, 1), np.float32)
    x[0, 2, 0] = 1.0
    k = np.arange(1, 5, dtype=np.float32).reshape(4, 1, 1)
    y = O.conv1d_same(x, k)[0, :, 0]
    # y[t] = sum_j k[j] x[t + j - 1]  (pad_left = 1): the impulse at 2 shows k[3] at t=0 ... k[0] at t=3
    assert y[3] == 1 and y[2] == 2 and y[1] == 3 and y[0] == 4 and y[4] == 0
    np.testing.assert_allclose(O.conv1d_same(x, k), TC.conv1d_same(x, k), atol=1e-7)


def test_cbhg_shapes_adjustment_and_lengths():
    d = V1_TINY
    w = init_weights_v1(d, 5, perturb=True)
    assert "v1/postnet/cbhg/adjustment/kernel" in w and "v1/encoder/cbhg/adjustment/kernel" in w   # 5 != 4, 5 != 3
    assert "v1/encoder/cbhg/adjustment/kernel" not in weight_shapes_v1(V1_PROFILE)                  # 128 == 256 // 2
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 6, d.encoder_prenet_out_units[1])).astype(np.float32)
    full = V.cbhg(w, "v1/encoder/cbhg", x, d.cbhg_out_units, d.max_filter_width, d.num_highway, None, d.bn_eps)
    assert full.shape == (2, 6, d.cbhg_out_units)
    part = V.cbhg(w, "v1/encoder/cbhg", x, d.cbhg_out_units, d.max_filter_width, d.num_highway, np.array([6, 4]), d.bn_eps)
    assert np.all(part[1, 4:] == 0) and np.allclose(part[0], full[0])
    half = d.cbhg_out_units // 2
    assert np.allclose(part[1, :4, :half], full[1, :4, :half])         # fw GRU of the short row is unaffected by its padding
    assert not np.allclose(part[1, :4, half:], full[1, :4, half:])     # bw GRU starts at len-1 instead of T-1


@pytest.mark.parametrize("seed", range(3))
def test_reference_property_on_the_v1_oracle(seed):
    """tests/decoder_test.py:42-67 restated on the oracle: free-run output fed back as training target."""
    rng = np.random.default_rng(seed)
    d = V1_TINY.replace(max_iters=int(rng.integers(3, 9)), min_iters=100)
    w = init_weights_v1(d, seed)
    B, T = int(rng.integers(1, 4)), int(rng.integers(2, 7))
    memory = rng.standard_normal((B, T, d.cbhg_out_units)).astype(np.float32)
    lens = np.full((B,), T)
    mel_i, stop_i, st_i = V.decoder_v1(d, w, memory, memory_sequence_length=lens)
    mel_t, stop_t, st_t = V.decoder_v1(d, w, memory, is_training=True, memory_sequence_length=lens, target=mel_i)
    np.testing.assert_allclose(mel_t, mel_i, atol=1e-6)
    np.testing.assert_allclose(stop_t, stop_i, atol=1e-6)
    np.testing.assert_allclose(np.stack(st_t[0].alignment_history), np.stack(st_i[0].alignment_history), atol=1e-6)


@pytest.mark.parametrize("name", ["v1_tiny_teacher", "v1_mid_teacher", "v1_mid_free"])
def test_v1_oracle_reproduces_committed_golden(name):
    import os
    from tests.golden import make_golden_v1 as G
    d, w, inp = G.materialize(name, G.cases()[name])
    gold = np.load(os.path.join(os.path.dirname(G.__file__), name + ".npz"))
    ref = G.run_oracle(d, w, inp)
    assert int(ref["steps"]) == int(gold["steps"])
    for k in ("memory", "mel", "stop", "alignment", "spec"):
        np.testing.assert_allclose(ref[k], gold[k], atol=1e-5, err_msg=k)


def test_postnet_training_oracle_forward_and_gradients():
    """oracle/v1_torch_train.py (the fp64 autograd checker of the HIP backward pass of TacotronV1PostNetModel's training step):
    (1) with the moving statistics in place of the batch statistics its forward is the inference PostNet of the NumPy oracle;
    (2) its gradients agree with central finite differences of its own loss on a sample of parameters of every layer kind."""
    from oracle import v1_torch_train as VT
    from tacotron2_amd.hparams import V1Dims
    from tacotron2_amd.weights_v1 import init_weights_v1
    d = V1Dims(num_mels=6, num_freq=9, post_net_cbhg_out_units=8, post_net_conv_channels=5, post_net_max_filter_width=4,
               post_net_projection1_out_channels=7, post_net_projection2_out_channels=6, post_net_num_highway=2)
    w = init_weights_v1(d, 3, perturb=True)
    rng = np.random.default_rng(1)
    B, T = 2, 7
    mel = rng.uniform(0, 1, (B, T, d.num_mels)).astype(np.float32)
    spec = rng.uniform(0, 1, (B, T, d.num_freq)).astype(np.float32)
    mask = (np.arange(T)[None, :] < np.array([7, 4])[:, None]).astype(np.float32)
    inf = VT.postnet_train_forward_backward(d, w, mel, spec, mask, use_moving=True)
    np.testing.assert_allclose(inf["spec"], V.postnet_v1(d, w, mel), atol=2e-5)
    ref = VT.postnet_train_forward_backward(d, w, mel, spec, mask)
    assert abs(ref["loss"] - float(np.sum(np.abs(ref["spec"] - spec) * mask[:, :, None]) / (mask.sum() * d.num_freq))) < 1e-6
    names = ["v1/postnet/cbhg/conv1d_K4/kernel", "v1/postnet/cbhg/conv1d_K2/batch_normalization/gamma", "v1/postnet/cbhg/proj1/kernel",
             "v1/postnet/cbhg/proj2/batch_normalization/beta", "v1/postnet/cbhg/adjustment/kernel", "v1/postnet/cbhg/highway_1/T/kernel",
             "v1/postnet/cbhg/highway_2/H/bias", "v1/postnet/cbhg/bidirectional_rnn/fw/gru_cell/gates/kernel",
             "v1/postnet/cbhg/bidirectional_rnn/bw/gru_cell/candidate/kernel", "v1/postnet/cbhg/bidirectional_rnn/bw/gru_cell/gates/bias",
             "v1/postnet/linear/kernel"]
    eps = 1e-4
    for name in names:
        flat = w[name].reshape(-1)
        for idx in rng.choice(flat.size, size=min(3, flat.size), replace=False):
            vals = []
            for sgn in (1, -1):
                w2 = dict(w)
                a = w[name].astype(np.float64).copy().reshape(-1)
                a[idx] += sgn * eps
                w2[name] = a.reshape(w[name].shape)
                vals.append(VT.postnet_train_forward_backward(d, w2, mel, spec, mask)["loss"])
            fd = (vals[0] - vals[1]) / (2 * eps)
            g = float(ref["grads"][name].reshape(-1)[idx])
            assert abs(fd - g) <= 2e-3 * max(abs(fd), abs(g)) + 2e-6, (name, idx, fd, g)
