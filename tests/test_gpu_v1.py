"""GPU parity of the Tacotron v1 path through the C-ABI (t2v1_*) against the NumPy
oracle (oracle/v1_oracle.py).  Tolerance: 1e-3 absolute on teacher-forced mel / stop / alignments and on the
encoder / PostNet outputs (north_star), identical stop-step index in free-running decode."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import t2_oracle as O
from oracle import v1_oracle as V
from tacotron2_amd.hparams import V1_PROFILE, V1_TINY
from tacotron2_amd.weights_v1 import init_weights_v1

ATOL = 1e-3
V1_MID = V1_PROFILE.replace(num_symbols=60, embedding_dim=48, encoder_prenet_out_units=(48, 40), cbhg_out_units=72, conv_channels=20,
                            max_filter_width=5, projection1_out_channels=36, projection2_out_channels=40, num_highway=2,
                            decoder_prenet_out_units=(44, 36), attention_out_units=136, decoder_out_units=100, num_mels=20,
                            num_freq=33, post_net_cbhg_out_units=24, post_net_conv_channels=12, post_net_max_filter_width=4,
                            post_net_projection1_out_channels=28, post_net_projection2_out_channels=20, post_net_num_highway=2,
                            max_iters=40, min_iters=5)
DIMS = {"tiny": V1_TINY, "mid": V1_MID, "ref": V1_PROFILE}
_engines = {}


def _engine(name, seed=7):
    from tacotron2_amd.tacotron_v1 import EngineV1
    key = (name, seed)
    if key not in _engines:
        d = DIMS[name]
        w = init_weights_v1(d, seed, perturb=True)
        _engines[key] = (d, w, EngineV1(d, w))
    return _engines[key]


def _np(t):
    return t.detach().cpu().numpy()


def _lens(rng, B, T):
    lens = rng.integers(max(1, T // 2), T + 1, size=B)
    lens[rng.integers(0, B)] = T
    return lens.astype(np.int64)


@pytest.mark.parametrize("name,B,T", [("tiny", 3, 7), ("tiny", 17, 9), ("mid", 5, 23), ("ref", 2, 19)])
@pytest.mark.parametrize("with_lens", [False, True])
def test_encoder_v1(name, B, T, with_lens):
    d, w, eng = _engine(name)
    rng = np.random.default_rng(B * 100 + T)
    ids = rng.integers(0, d.num_symbols, (B, T))
    lens = _lens(rng, B, T) if with_lens else None
    ref = V.encoder_v1(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    np.testing.assert_allclose(_np(eng.encode(ids, lens)), ref, atol=ATOL)
    emb = O.embedding_lookup(ids, w["embedding"])
    np.testing.assert_allclose(_np(eng.encode_embedded(emb, lens)), ref, atol=ATOL)


@pytest.mark.parametrize("B,T", [(20, 37), (1, 5), (33, 12)])
def test_cbhg_whole_sequence_gru(B, T):
    """The CBHG BiGRU of 128 units per direction as one persistent launch (gru_seq_kernel, option gruseq) against the
    oracle and against the per-step launches, with ragged lengths (state copied through, zero output past the end,
    per-row reversed positions for the backward direction) and partially filled 16-row tiles."""
    d, w, eng = _engine("ref")
    assert d.post_net_cbhg_out_units // 2 == 128
    rng = np.random.default_rng(B + T)
    mel = rng.uniform(0, 1, (B, T, d.num_mels)).astype(np.float32)
    lens = _lens(rng, B, T)
    ref = V.cbhg(w, "v1/postnet/cbhg", mel, d.post_net_cbhg_out_units, d.post_net_max_filter_width, d.post_net_num_highway, lens, d.bn_eps)
    got = _np(eng.cbhg(1, mel, lens))
    np.testing.assert_allclose(got, ref, atol=ATOL)
    eng.set_option("gruseq", 0)
    steps = _np(eng.cbhg(1, mel, lens))
    eng.set_option("gruseq", 1)
    np.testing.assert_allclose(got, steps, atol=1e-5)
    for b in range(B):
        assert not got[b, lens[b]:].any()


def test_embedding_range_and_shape_errors():
    from tacotron2_amd._lib import T2Error
    d, w, eng = _engine("tiny")
    with pytest.raises(T2Error) as ei:
        eng.encode(np.full((1, 3), d.num_symbols), None)
    assert ei.value.code == -5
    with pytest.raises(T2Error):
        eng.encode(np.zeros((65, 3), np.int64), None)              # one reference batch <= T2_MAX_BATCH


@pytest.mark.parametrize("name,B,T", [("tiny", 3, 8), ("mid", 4, 21), ("ref", 2, 16)])
def test_cbhg_and_postnet(name, B, T):
    d, w, eng = _engine(name)
    rng = np.random.default_rng(T)
    mel = rng.uniform(0, 1, (B, T, d.num_mels)).astype(np.float32)
    np.testing.assert_allclose(_np(eng.postnet(mel)), V.postnet_v1(d, w, mel), atol=ATOL)
    lens = _lens(rng, B, T)
    ref = V.cbhg(w, "v1/postnet/cbhg", mel, d.post_net_cbhg_out_units, d.post_net_max_filter_width, d.post_net_num_highway, lens, d.bn_eps)
    np.testing.assert_allclose(_np(eng.cbhg(1, mel, lens)), ref, atol=ATOL)
    x = rng.standard_normal((B, T, d.encoder_prenet_out_units[1])).astype(np.float32)
    ref = V.cbhg(w, "v1/encoder/cbhg", x, d.cbhg_out_units, d.max_filter_width, d.num_highway, None, d.bn_eps)
    np.testing.assert_allclose(_np(eng.cbhg(0, x, None)), ref, atol=ATOL)


@pytest.mark.parametrize("name,B,T,S", [("tiny", 3, 7, 6), ("tiny", 20, 9, 5), ("mid", 5, 17, 19), ("ref", 2, 30, 24), ("ref", 33, 12, 6),
                                        ("ref", 2, 1300, 3)])          # long input: the attention kernel's chunked sweep (6 chunks, ragged last)
@pytest.mark.parametrize("teacher_forcing", [True, False])
def test_decoder_v1_teacher_and_validation(name, B, T, S, teacher_forcing):
    d, w, eng = _engine(name)
    rng = np.random.default_rng(B + T + S)
    memory = rng.standard_normal((B, T, d.cbhg_out_units)).astype(np.float32) * 0.5
    lens = _lens(rng, B, T)
    target = rng.uniform(0, 1, (B, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    mel_o, stop_o, st = V.decoder_v1(d, w, memory, is_validation=True, teacher_forcing=teacher_forcing,
                                     memory_sequence_length=lens, target=target, max_iters=S)
    n0 = eng.get_stat("train_persist_launches")
    mel, stop, align = eng.decode_teacher(memory, lens, target, teacher_forcing=teacher_forcing)
    # teacher-forced decodes at the reference widths run as the training forward's ONE persistent launch (csrc/v1_dec_train.h)
    assert eng.get_stat("train_persist_launches") == n0 + (1 if teacher_forcing and name == "ref" else 0)
    np.testing.assert_allclose(_np(mel), mel_o, atol=ATOL)
    np.testing.assert_allclose(_np(stop), stop_o, atol=ATOL)
    np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=ATOL)
    if teacher_forcing:          # the per-step PreNet / projection launches (option tfbatch = 0: launch-per-phase schedule) give the same frames
        eng.set_option("tfbatch", 0)
        mel2, stop2, align2 = eng.decode_teacher(memory, lens, target, teacher_forcing=True)
        assert eng.get_stat("train_persist_launches") == n0 + (1 if name == "ref" else 0)
        eng.set_option("tfbatch", 1)
        np.testing.assert_allclose(_np(mel2), mel_o, atol=ATOL)
        np.testing.assert_allclose(_np(mel2), _np(mel), atol=1e-5)
        np.testing.assert_allclose(_np(stop2), _np(stop), atol=1e-5)
        np.testing.assert_allclose(_np(align2), _np(align), atol=1e-6)


def _stop_tuned(d, w, memory, lens, lo, hi, min_gap=4e-3):
    """Threshold in the widest gap of the per-step batch-minimum stop logits (either sign of the stop kernel)."""
    best = None
    for sign in (1.0, -1.0):
        w2 = dict(w)
        w2["v1/decoder/stop_token_projection/kernel"] = (sign * w["v1/decoder/stop_token_projection/kernel"]).astype(np.float32)
        w2["v1/decoder/stop_token_projection/bias"] = np.zeros(1, np.float32)
        _, stop, _ = V.decoder_v1(d.replace(min_iters=hi + 1), w2, memory, memory_sequence_length=lens, max_iters=hi)
        m = stop[:, :, 0].min(axis=0)
        window = m[d.min_iters + 1:hi]
        srt = np.sort(window)
        for a, b in zip(srt[:-1], srt[1:]):
            if b - a < min_gap:
                continue
            thr = (a + b) / 2
            t = d.min_iters + 1 + int(np.argmax(window > thr))
            if lo <= t < hi - 1 and (best is None or b - a > best[1]):
                best = (t, b - a, thr, w2)
    assert best is not None, "no usable threshold"
    best[3]["v1/decoder/stop_token_projection/bias"] = np.array([-best[2]], np.float32)
    return best[3], best[0] + 1


@pytest.mark.parametrize("B,T", [(1, 11), (6, 14)])
def test_v1_free_run_stop_index_identical(B, T):
    from tacotron2_amd.tacotron_v1 import DecoderV1, EngineV1
    d = V1_MID
    w0 = init_weights_v1(d, 21, perturb=True)
    rng = np.random.default_rng(B)
    memory = rng.standard_normal((B, T, d.cbhg_out_units)).astype(np.float32) * 0.5
    lens = _lens(rng, B, T)
    w, steps_expected = _stop_tuned(d, w0, memory, lens, 8, 30)
    mel_o, stop_o, st = V.decoder_v1(d, w, memory, memory_sequence_length=lens)
    assert stop_o.shape[1] == steps_expected
    eng = EngineV1(d, w)
    dec = DecoderV1(d.decoder_prenet_out_units, 0.5, d.attention_out_units, d.decoder_out_units, d.num_mels, d.outputs_per_step,
                    d.max_iters, d.n_feed_frame, min_iters=d.min_iters, engine=eng)
    mel, stop, state = dec(memory, is_training=False, is_validation=False, memory_sequence_length=lens)
    assert stop.shape[1] == steps_expected
    np.testing.assert_allclose(_np(mel), mel_o, atol=ATOL)
    np.testing.assert_allclose(_np(stop), stop_o, atol=ATOL)
    np.testing.assert_allclose(_np(state[0].alignment_history.stack()), np.stack(st[0].alignment_history, 0), atol=ATOL)
    eng.close()


@pytest.mark.parametrize("seed", range(4))
def test_reference_property_v1(seed):
    """The reference's own test (tests/decoder_test.py:42-67): DecoderV1(drop_rate=0) free-running output fed back
    as the training target reproduces alignments, mel and stop token."""
    from tacotron2_amd.tacotron_v1 import DecoderV1, EngineV1
    rng = np.random.default_rng(seed)
    bs, num_mels, r = int(rng.integers(1, 4)), int(rng.integers(2, 21)), int(rng.integers(1, 5))
    T_in, max_iters = int(rng.integers(2, 9)), int(rng.integers(3, 13))
    d = V1_PROFILE.replace(num_mels=num_mels, outputs_per_step=r, n_feed_frame=int(rng.integers(1, r + 1)), max_iters=max_iters,
                           min_iters=max_iters + 1, post_net_projection2_out_channels=num_mels)
    eng = EngineV1(d, init_weights_v1(d, seed))
    dec = DecoderV1(d.decoder_prenet_out_units, 0.0, d.attention_out_units, d.decoder_out_units, num_mels, r, max_iters,
                    d.n_feed_frame, min_iters=d.min_iters, engine=eng)
    memory = rng.standard_normal((bs, T_in, d.cbhg_out_units)).astype(np.float32)
    lens = np.full((bs,), T_in, np.int64)
    out_i, stop_i, st_i = dec(memory, is_training=False, is_validation=False, memory_sequence_length=lens)
    out_t, stop_t, st_t = dec(memory, is_training=True, is_validation=False, memory_sequence_length=lens, target=out_i)
    np.testing.assert_allclose(_np(st_t[0].alignment_history.stack()), _np(st_i[0].alignment_history.stack()), atol=1e-5)
    np.testing.assert_allclose(_np(out_t), _np(out_i), atol=1e-5)
    np.testing.assert_allclose(_np(stop_t), _np(stop_i), atol=1e-5)
    eng.close()


def test_v1_model_fn_predict_and_eval():
    from tacotron2_amd.tacotron_v1 import ModeKeys, SingleSpeakerTacotronV1Model, TacotronV1PostNetModel
    d, w, eng = _engine("mid")
    rng = np.random.default_rng(5)
    B, T, S = 3, 9, 7
    ids = rng.integers(0, d.num_symbols, (B, T))
    lens = _lens(rng, B, T)
    model = SingleSpeakerTacotronV1Model(d, engine=eng)
    target = rng.uniform(0, 1, (B, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    labels = {"mel": target, "done": np.zeros((B, S), np.float32), "spec_loss_mask": np.ones((B, S * d.outputs_per_step), np.float32),
              "binary_loss_mask": np.ones((B, S), np.float32)}
    ev = model.model_fn({"source": ids, "source_length": lens}, labels, ModeKeys.EVAL)
    ref = V.tacotron1_forward(d, w, ids, lens, mode="eval", target=target, teacher_forcing=True)
    np.testing.assert_allclose(float(ev["eval_metric_ops"]["mel_loss_with_teacher"]),
                               float(O.spec_loss(ref["mel"], target, labels["spec_loss_mask"])), atol=1e-4)
    ref = V.tacotron1_forward(d, w, ids, lens, mode="eval", target=target, teacher_forcing=False)
    np.testing.assert_allclose(_np(ev["mel"]), ref["mel"], atol=ATOL)
    np.testing.assert_allclose(_np(ev["alignment"]), ref["alignment"], atol=ATOL)
    post = TacotronV1PostNetModel(d, eng)
    spec = post.model_fn({"mel": ref["mel"]}, None, ModeKeys.PREDICT)["predictions"]["spec"]
    np.testing.assert_allclose(_np(spec), V.postnet_v1(d, w, ref["mel"]), atol=ATOL)
    with pytest.raises(ValueError):        # TRAIN needs the variables' initial values (weights=...); the step itself: tests/test_gpu_v1_train.py
        model.model_fn({"source": ids, "source_length": lens}, labels, ModeKeys.TRAIN)


def test_v1_constructor_errors():
    from tacotron2_amd.tacotron_v1 import CBHG, DecoderV1
    with pytest.raises(ValueError):
        DecoderV1(decoder_out_units=0)                                 # rnn_wrappers.py:155-156
    with pytest.raises(AssertionError):
        CBHG(7, 4, 4, 4, 4, 2, False)                                  # modules.py:169
    with pytest.raises(AssertionError):
        V1_PROFILE.replace(n_feed_frame=3)                             # helpers.py:55


@pytest.mark.parametrize("name", ["v1_tiny_teacher", "v1_mid_teacher", "v1_mid_free", "v1_ref_teacher"])
def test_v1_golden_fixtures_through_the_api(name):
    """Committed fixtures (tests/golden/make_golden_v1.py) through the reference-shaped classes: encoder memory,
    teacher-forced / free-running mel, stop, alignments, PostNet spectrogram; identical stop-step index."""
    import os
    from tests.golden import make_golden_v1 as G
    from tacotron2_amd.tacotron_v1 import EngineV1, ModeKeys, SingleSpeakerTacotronV1Model, TacotronV1PostNetModel
    d, w, inp = G.materialize(name, G.cases()[name])
    gold = np.load(os.path.join(os.path.dirname(G.__file__), name + ".npz"))
    assert abs(G.weights_checksum(w) - float(gold["weights_checksum"])) < 1e-3 * float(gold["weights_checksum"])
    eng = EngineV1(d, w)
    model = SingleSpeakerTacotronV1Model(d, engine=eng)
    memory = eng.encode(inp["ids"], None)
    if inp["mode"] == "teacher":
        mel, stop, st = model.decoder(memory, is_training=False, is_validation=True, teacher_forcing=True,
                                      memory_sequence_length=inp["lens"], target=inp["targets"])
        align = st[0].alignment_history.stack().permute(1, 2, 0)
    else:
        pred = model.predict({"source": inp["ids"], "source_length": inp["lens"]})
        mel, stop, align = pred["mel"], pred["stop_token"], pred["alignment"]
        assert stop.shape[1] == int(gold["steps"])
    spec = TacotronV1PostNetModel(d, eng).model_fn({"mel": mel}, None, ModeKeys.PREDICT)["predictions"]["spec"]
    for k, got in (("memory", memory), ("mel", mel), ("stop", stop), ("alignment", align), ("spec", spec)):
        np.testing.assert_allclose(_np(got), gold[k], atol=ATOL, err_msg=f"{name}:{k}")
    eng.close()
