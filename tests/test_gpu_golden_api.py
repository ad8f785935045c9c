"""HIP path against the committed golden fixtures (tests/golden/*.npz, made by the CPU oracle) and
through the reference-shaped Python API (tacotron_v2.py / rnn_wrappers.py / helpers.py mirrors)."""
import os

import numpy as np
import pytest
import torch

from oracle import t2_oracle as O
from tacotron2_amd.hparams import PROFILE_TINY
from tacotron2_amd.weights import init_weights
from tests.golden import make_golden as G

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATOL = 2e-4          # north_star bound is 1e-3 abs


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", ["tiny_r1_nf1", "tiny_r2_nf1", "tiny_r2_nf2", "mid_teacher", "mid_free", "paper_b1_teacher",
                                  "paper_sharp_b1_teacher", "paper_sharp_b64_free"])
def test_hip_matches_golden(name):
    from tacotron2_amd.tacotron_v2 import TacotronV2Model, ModeKeys
    d, w, inp = G.materialize(name, G.cases()[name])
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    assert abs(G.weights_checksum(w) - float(gold["weights_checksum"])) < 1e-3 * float(gold["weights_checksum"])
    model = TacotronV2Model(d, w)
    feats = {"id": None, "text": None, "source": inp["ids"], "source_length": inp["lens"]}
    if inp["mode"] == "teacher":
        B, T = inp["targets"].shape[:2]
        S = T // d.outputs_per_step
        labels = {"mel": inp["targets"], "done": np.zeros((B, S), np.float32),
                  "spec_loss_mask": np.ones((B, T), np.float32), "binary_loss_mask": np.ones((B, S), np.float32)}
        memory = model.engine.encode(inp["ids"], inp["lens"])
        mel, stop, st = model.decoder(memory, is_training=False, is_validation=True, teacher_forcing=True,
                                      memory_sequence_length=inp["lens"], target=inp["targets"])
        align = st[0].alignment_history.stack().permute(1, 2, 0)
        post = model.postnet(mel)
        ev = model.model_fn(feats, labels, ModeKeys.EVAL)
        ref_loss = O.spec_loss(gold["mel"], inp["targets"], labels["spec_loss_mask"])
        assert abs(float(ev["eval_metric_ops"]["mel_loss_with_teacher"]) - float(ref_loss)) < 1e-4
    else:
        pred = model.predict(feats)
        memory = model.engine.encode(inp["ids"], inp["lens"])
        mel, stop, align, post = pred["mel"], pred["stop_token"], pred["alignment"], pred["mel_postnet"]
        assert stop.shape[1] == int(gold["steps"])          # identical stop-step index
    if "memory_rows" in gold.files:                          # large batches keep a sample of the encoder output rows
        memory = memory[torch.as_tensor(gold["memory_rows"], device=memory.device)]
    if "sharp" in name:
        # the fixture must be able to fail: a peaked, moving attention.  Such a recurrence has windows in which rounding noise is
        # amplified (the cumulative-state feedback) and decays again; the fixture records the oracle's OWN fp32-vs-fp64 difference
        # per decoder step (make_golden.fp64_noise: up to 8.6e-4 in one window of the 400-step case, ~5e-6 elsewhere) and the
        # tolerance of a step is ATOL + 4 x that floor.  Measured on MI355X: the HIP error is 1.1 .. 1.7 x the floor in the window.
        peak = np.transpose(gold["alignment"], (2, 0, 1)).max(-1)
        assert peak.max() > 0.8 and peak.mean() > 0.4 and len(set(gold["alignment"][0].argmax(0).tolist())) >= 3
        S = int(gold["steps"])
        err = np.abs(_np(align) - gold["alignment"]).max((0, 1))
        assert (err <= ATOL + 4 * gold["alignment_noise"]).all(), (name, "alignment", float(err.max()))
        err = np.abs(_np(mel) - gold["mel"]).reshape(mel.shape[0], S, -1).max((0, 2))
        assert (err <= ATOL + 4 * gold["mel_noise"]).all(), (name, "mel", float(err.max()))
        err = np.abs(_np(stop) - gold["stop"]).reshape(stop.shape[0], S).max(0)
        assert (err <= ATOL + 4 * gold["stop_noise"]).all(), (name, "stop", float(err.max()))
        np.testing.assert_allclose(_np(memory), gold["memory"], atol=ATOL, err_msg=f"{name}:memory")
        np.testing.assert_allclose(_np(post), gold["mel_postnet"], atol=ATOL, err_msg=f"{name}:mel_postnet")
    else:
        for k, got in (("memory", memory), ("mel", mel), ("stop", stop), ("alignment", align), ("mel_postnet", post)):
            np.testing.assert_allclose(_np(got), gold[k], atol=ATOL, err_msg=f"{name}:{k}")
    model.close()


@pytest.mark.parametrize("seed", range(4))
def test_reference_property_free_run_equals_teacher_forced(seed):
    """The reference's own test (tests/decoder_test.py:42-67) on the HIP decoder, V2 cells:
    free-running output fed back as teacher-forcing target reproduces alignments, mel and stop."""
    from tacotron2_amd.tacotron_v2 import DecoderV2
    rng = np.random.default_rng(seed)
    bs, num_mels, r = int(rng.integers(1, 4)), int(rng.integers(2, 21)), int(rng.integers(1, 5))
    t_in, depth = int(rng.integers(5, 21)), int(rng.integers(2, 11)) * 2
    memory = rng.uniform(-1, 1, (bs, t_in, depth)).astype(np.float32)
    lens = np.repeat(t_in, bs)
    dec = DecoderV2(prenet_out_units=(16, 8), drop_rate=0.0, attention_out_units=12, decoder_out_units=20, num_mels=num_mels,
                    outputs_per_step=r, max_iters=10, n_feed_frame=1, attention_dim=8, attention_kernel=5, attention_filters=4,
                    seed=seed)
    out_i, stop_i, st_i = dec(torch.from_numpy(memory).cuda(), is_training=False, memory_sequence_length=lens)
    assert stop_i.shape[1] == 10
    out_t, stop_t, st_t = dec(torch.from_numpy(memory).cuda(), is_training=True, memory_sequence_length=lens, target=out_i)
    al_i = st_i[0].alignment_history.stack().permute(1, 2, 0)
    al_t = st_t[0].alignment_history.stack().permute(1, 2, 0)
    np.testing.assert_allclose(_np(al_i), _np(al_t), atol=1e-6)
    np.testing.assert_allclose(_np(out_i), _np(out_t), atol=1e-6)
    np.testing.assert_allclose(_np(stop_i), _np(stop_t), atol=1e-6)


def test_generic_helper_loop_equals_fused_loops_and_oracle():
    """RNNCell / Helper protocol path (one t2_decode_step per iteration, helpers.py rules on the
    host) against the fused device loops and the oracle."""
    from tacotron2_amd.tacotron_v2 import DecoderV2
    d = G.MID
    w = init_weights(d, 9, perturb=True)
    rng = np.random.default_rng(5)
    B, T = 3, 14
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    lens = np.array([14, 9, 12])
    w, steps = G.tune_stop(d, w, memory, lens, 15, 36)
    dec = DecoderV2(prenet_out_units=d.prenet_units, drop_rate=0.0, attention_out_units=d.attention_lstm_units,
                    decoder_out_units=d.decoder_lstm_units, num_mels=d.num_mels, outputs_per_step=d.outputs_per_step,
                    max_iters=d.max_iters, n_feed_frame=d.n_feed_frame, attention_dim=d.attention_dim,
                    attention_kernel=d.attention_kernel, attention_filters=d.attention_filters,
                    zoneout_factor_cell=d.zoneout_cell, zoneout_factor_output=d.zoneout_output, weights=w)
    mem_t = torch.from_numpy(memory).cuda()
    mel_o, stop_o, st_o = O.decoder_v2(d, w, memory, memory_sequence_length=lens)
    assert stop_o.shape[1] == steps
    for generic in (False, True):
        mel, stop, st = dec(mem_t, is_training=False, memory_sequence_length=lens, generic_loop=generic)
        assert stop.shape[1] == steps
        np.testing.assert_allclose(_np(mel), mel_o, atol=5 * ATOL)
        np.testing.assert_allclose(_np(stop), stop_o, atol=5 * ATOL)
        np.testing.assert_allclose(_np(st[0].alignment_history.stack()), np.stack(st_o[0].alignment_history), atol=5 * ATOL)
    tgt = rng.uniform(0, 1, (B, 12, d.num_mels)).astype(np.float32)
    mel_o, stop_o, _ = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=tgt)
    for generic in (False, True):
        mel, stop, _ = dec(mem_t, is_training=False, is_validation=True, teacher_forcing=True, memory_sequence_length=lens,
                           target=tgt, generic_loop=generic)
        np.testing.assert_allclose(_np(mel), mel_o, atol=ATOL)
        np.testing.assert_allclose(_np(stop), stop_o, atol=ATOL)


def test_api_rejections():
    from tacotron2_amd.tacotron_v2 import LocationSensitiveAttention, DecoderV2, EncoderV2
    mem = torch.zeros(1, 4, 6)
    with pytest.raises(TypeError):
        LocationSensitiveAttention(8, mem, None, 5, 3, smoothing=True)
    with pytest.raises(AssertionError):
        EncoderV2(1, 3, 7, 0.0, 0.0, 0.0, False)
    dec = DecoderV2(prenet_out_units=(8,), drop_rate=0.5, attention_out_units=8, decoder_out_units=8, num_mels=4,
                    outputs_per_step=2, attention_dim=4, attention_kernel=3, attention_filters=2)
    with pytest.raises(AssertionError):
        DecoderV2(num_mels=4, outputs_per_step=1, n_feed_frame=2)(mem.cuda(), is_training=False)


def test_synthesize_driver_and_splitk_option():
    from tacotron2_amd.synthesize import synthesize
    from tacotron2_amd.tacotron_v2 import TacotronV2Model
    d = G.MID.replace(num_symbols=128, max_iters=14)
    w = init_weights(d, 3, perturb=True)
    model = TacotronV2Model(d, w)
    texts = ["HELLO WORLD", "A", "MI355X SPEAKS"] * 7            # 21 rows -> MB = 2
    res = synthesize(model, texts)
    assert len(res) == 21 and res[0]["mel"].shape[1] == d.num_mels and res[1]["alignment"].shape[0] == 2
    model.engine.set_option("splitk", 1)                         # split-K LSTM step must give the same frames
    res2 = synthesize(model, texts)
    for a, b in zip(res, res2):
        np.testing.assert_allclose(a["mel"], b["mel"], atol=1e-5)
    ids, lens = np.array([[72, 73, 1]]), np.array([3])
    ref = O.tacotron2_forward(d, w, ids, lens, mode="predict")
    got = synthesize(model, ["HI"])[0]
    np.testing.assert_allclose(got["mel"], ref["mel_postnet"][0], atol=5 * ATOL)
    model.close()


def test_inference_helper_drives_the_hip_decoder():
    """H4, the legacy InferenceHelper (helpers.py:13-49: a row finishes when its whole output is exactly 0.0) through the
    generic RNNCell / Helper loop on the HIP cell stack, against the oracle's dynamic_decode with the same helper.  A row
    whose projection kernels and biases are zero... does not exist per row, so the zero output is forced for EVERY row
    from the start: all-zero projection weights end the decode after one step; non-zero ones run to maximum_iterations."""
    from tacotron2_amd.helpers import InferenceHelper
    from tacotron2_amd.modules import PreNet, ZoneoutLSTMCell
    from tacotron2_amd.rnn_wrappers import AttentionRNN, OutputAndStopTokenWrapper
    from tacotron2_amd.tacotron_v2 import DecoderRNNV2, LocationSensitiveAttention, dynamic_decode
    d = PROFILE_TINY
    rng = np.random.default_rng(4)
    B, T = 3, 6
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    lens = np.array([6, 4, 5])
    for zero_projection, want_steps in ((False, 5), (True, 1)):
        w = init_weights(d, 9, perturb=True)
        if zero_projection:
            for k in ("decoder/out_projection/kernel", "decoder/out_projection/bias"):
                w[k] = np.zeros_like(w[k])
        mech = LocationSensitiveAttention(d.attention_dim, memory, lens, d.attention_kernel, d.attention_filters)
        cell = OutputAndStopTokenWrapper(
            DecoderRNNV2(d.decoder_lstm_units,
                         AttentionRNN(ZoneoutLSTMCell(d.attention_lstm_units, False, d.zoneout_cell, d.zoneout_output),
                                      tuple(PreNet(u, False) for u in d.prenet_units), mech),
                         False, d.zoneout_cell, d.zoneout_output),
            d.num_mels * d.outputs_per_step, weights=w, n_feed_units=d.num_mels * d.n_feed_frame, num_mels=d.num_mels)
        helper = InferenceHelper(B, d.num_mels, d.outputs_per_step, n_feed_frame=d.n_feed_frame)
        (mel, stop), state, steps = dynamic_decode(cell, helper, cell.zero_state(B), maximum_iterations=5)
        assert steps == want_steps
        ocell = O.DecoderCellV2(d, w, memory, lens)
        (mel_o, stop_o), _, steps_o = O.dynamic_decode(ocell, O.InferenceHelper(B, d.num_mels, d.outputs_per_step, d.n_feed_frame),
                                                       ocell.zero_state(), 5)
        assert steps_o == want_steps
        np.testing.assert_allclose(_np(mel), mel_o, atol=ATOL)
        np.testing.assert_allclose(_np(stop), stop_o, atol=ATOL)
        cell.engine().close()


def test_imported_checkpoint_runs_on_the_engine(tmp_path):
    """SURVEY.md 8f rank 2 on the GPU: TF-style variable dict -> import_tf_variables -> save_npz -> load_npz -> Engine, against
    the oracle on the same dict."""
    from tacotron2_amd import checkpoint as C
    from tacotron2_amd.engine import Engine
    d = PROFILE_TINY
    w = init_weights(d, 31, perturb=True)
    names = C.tf_style_names(d)
    tf_vars = {names[k] + ":0": v for k, v in w.items()}
    tf_vars["global_step"] = np.array(3, np.int64)
    tf_vars[names["embedding"] + "/Adam"] = np.zeros_like(w["embedding"])
    path = str(tmp_path / "ckpt.npz")
    C.save_npz(path, C.import_tf_variables(tf_vars, d))
    loaded = C.load_npz(path, d)
    eng = Engine(d, loaded)
    rng = np.random.default_rng(2)
    ids = rng.integers(0, d.num_symbols, (4, 9))
    lens = np.array([9, 5, 7, 3])
    ref = O.tacotron2_forward(d, w, ids, lens, mode="predict")
    memory = eng.encode(ids, lens)
    mel, stop, align, n = eng.decode_free(memory, lens)
    assert n == ref["steps"]
    np.testing.assert_allclose(_np(memory), ref["memory"], atol=ATOL)
    np.testing.assert_allclose(_np(mel), ref["mel"], atol=ATOL)
    np.testing.assert_allclose(_np(eng.postnet(mel)), ref["mel_postnet"], atol=ATOL)
    eng.close()


def test_synthesize_sharded_covers_the_job_without_overlap():
    """SURVEY.md 8e, inference partitioning: two ranks (here: two calls) deal the length-sorted batches round robin; together
    they cover every utterance exactly once and each result equals the unsharded synthesis of the same batch."""
    from tacotron2_amd.sharding import make_batches
    from tacotron2_amd.synthesize import synthesize, synthesize_sharded, text_to_sequence
    from tacotron2_amd.tacotron_v2 import TacotronV2Model
    d = PROFILE_TINY.replace(num_symbols=128, max_iters=8)
    model = TacotronV2Model(d, init_weights(d, 3, perturb=True))
    texts = ["AB", "HELLO WORLD", "ABC", "TACOTRON", "X", "SPEECH SYNTH", "MI355X"]
    parts = [synthesize_sharded(model, texts, world_size=2, rank=r, batch_size=2) for r in range(2)]
    assert not set(parts[0]) & set(parts[1]) and set(parts[0]) | set(parts[1]) == set(range(len(texts)))
    batches = make_batches([len(text_to_sequence(t)) for t in texts], 2)
    for b, idx in enumerate(batches):
        whole = synthesize(model, [texts[i] for i in idx])
        for i, res in zip(idx, whole):
            np.testing.assert_array_equal(parts[b % 2][int(i)]["mel"], res["mel"])
    model.close()


@pytest.mark.parametrize("drop_rate,zoneout", [(0.5, 0.1), (0.5, 0.0), (0.0, 0.1)])
def test_single_cell_calls_with_sampling_equal_the_fused_training_forward_and_the_oracle(drop_rate, zoneout):
    """VERDICT round 2, missing #3: the RNNCell protocol with the training-mode sampling of the reference
    (DecoderPreNetWrapper.call rnn_wrappers.py:51-53 -> PreNet dropout modules.py:90,96-98; ZoneoutLSTMCell's sampled branch
    modules.py:274-276), one t2_decode_step_train per iteration of the generic Helper loop, against DecoderV2.call's fused
    training forward with the same seed (same counter-based masks) and against the oracle's sampled decoder."""
    from tacotron2_amd.tacotron_v2 import DecoderV2
    d = G.MID.replace(zoneout_cell=zoneout, zoneout_output=zoneout)
    w = init_weights(d, 9, perturb=True)
    rng = np.random.default_rng(5)
    B, T, S = 4, 13, 6
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    lens = np.array([13, 9, 12, 5])
    tgt = rng.uniform(0, 1, (B, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    seed = 77
    mel_o, stop_o, st_o = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=tgt, train_seed=seed,
                                       prenet_drop_rate=drop_rate)
    outs = []
    for generic in (True, False):
        dec = DecoderV2(prenet_out_units=d.prenet_units, drop_rate=drop_rate, attention_out_units=d.attention_lstm_units,
                        decoder_out_units=d.decoder_lstm_units, num_mels=d.num_mels, outputs_per_step=d.outputs_per_step,
                        max_iters=d.max_iters, n_feed_frame=d.n_feed_frame, attention_dim=d.attention_dim,
                        attention_kernel=d.attention_kernel, attention_filters=d.attention_filters,
                        zoneout_factor_cell=d.zoneout_cell, zoneout_factor_output=d.zoneout_output, weights=w, seed=seed)
        mel, stop, st = dec(torch.from_numpy(memory).cuda(), is_training=True, memory_sequence_length=lens, target=tgt, generic_loop=generic)
        assert dec.seed == seed + 1                                  # the mask seed advances per training-mode call
        outs.append((_np(mel), _np(stop), _np(st[0].alignment_history.stack())))
        np.testing.assert_allclose(outs[-1][0], mel_o, atol=ATOL)
        np.testing.assert_allclose(outs[-1][1], stop_o, atol=ATOL)
        np.testing.assert_allclose(outs[-1][2], np.stack(st_o[0].alignment_history), atol=ATOL)
    for a, b in zip(*outs):
        np.testing.assert_allclose(a, b, atol=1e-5)
    mel_i, _, _ = O.decoder_v2(d, w, memory, is_validation=True, teacher_forcing=True, memory_sequence_length=lens, target=tgt)
    assert np.abs(mel_i - mel_o).max() > 1e-3                        # the sampled forward is not the inference forward
