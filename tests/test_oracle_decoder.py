"""The reference's only test (tests/decoder_test.py:42-67: free-running output fed back as the
teacher-forcing target reproduces alignments, mel and stop token) restated for the V2 oracle,
plus the helper / dynamic_decode quirks of SURVEY.md §8a H1-H5 and Appendix B."""
import numpy as np
import pytest
from hypothesis import given, settings, HealthCheck
from hypothesis.strategies import integers, composite

from oracle import t2_oracle as O
from tacotron2_amd.hparams import PROFILE_TINY
from tacotron2_amd.weights import init_weights


@composite
def all_args(draw):
    bs = draw(integers(1, 3))
    num_mels = draw(integers(2, 20))
    r = draw(integers(1, 4))
    nf = draw(integers(1, r))
    t_in = draw(integers(5, 20))
    depth = draw(integers(2, 10)) * 2
    seed = draw(integers(0, 2 ** 16))
    return bs, num_mels, r, nf, t_in, depth, seed


@given(args=all_args())
@settings(max_examples=10, deadline=None, suppress_health_check=[HealthCheck.too_slow])
def test_decoder_free_run_equals_teacher_forced(args):
    bs, num_mels, r, nf, t_in, depth, seed = args
    d = PROFILE_TINY.replace(num_mels=num_mels, outputs_per_step=r, n_feed_frame=nf, enc_units=depth,
                             max_iters=10, min_iters=10)
    w = init_weights(d, seed, perturb=True)
    memory = np.random.default_rng(seed).uniform(-1, 1, (bs, t_in, depth)).astype(np.float32)
    lens = np.full((bs,), t_in)
    out_i, stop_i, st_i = O.decoder_v2(d, w, memory, is_training=False, memory_sequence_length=lens)
    assert stop_i.shape[1] == 10            # min_iters=10 => time > 10 never true within 10 iterations
    out_t, stop_t, st_t = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=out_i)
    np.testing.assert_allclose(O.alignment_from_state(st_i), O.alignment_from_state(st_t), atol=1e-6)
    np.testing.assert_allclose(out_i, out_t, atol=1e-6)
    np.testing.assert_allclose(stop_i, stop_t, atol=1e-6)
    # ValidationHelper, both flavours (helpers.py:131-138)
    out_v, stop_v, _ = O.decoder_v2(d, w, memory, is_validation=True, teacher_forcing=False,
                                    memory_sequence_length=lens, target=out_i)
    np.testing.assert_allclose(out_i, out_v, atol=1e-6)
    out_v2, _, _ = O.decoder_v2(d, w, memory, is_validation=True, teacher_forcing=True,
                                memory_sequence_length=lens, target=out_i)
    np.testing.assert_allclose(out_i, out_v2, atol=1e-6)


def _cell_with_stop_script(script, B, M, r):
    """A fake cell whose stop logits follow ``script[t]`` (list of per-row logits)."""
    def cell(inputs, state):
        t = state
        mel = np.full((B, M * r), float(t + 1), np.float32)
        stop = np.asarray(script[t], np.float32).reshape(B, 1)
        return (mel, stop), t + 1
    return cell


def test_stop_rule_is_batch_coupled_and_needs_time_gt_min_iters():
    """helpers.py:90-94: finished = reduce_all(sigmoid(stop) > 0.5 and time > min_iters, axis=0);
    the output of the stopping step is kept (App. A7)."""
    B, M, r = 2, 3, 2
    pos, neg = 5.0, -5.0
    script = [[pos, pos]] * 4 + [[pos, neg]] * 3 + [[neg, pos]] + [[pos, pos]] * 10
    helper = O.StopTokenBasedInferenceHelper(B, M, r, n_feed_frame=1, min_iters=2)
    (mel, stop), _, steps = O.dynamic_decode(_cell_with_stop_script(script, B, M, r), helper, 0, 50)
    # time 0..2 blocked by min_iters, time 3 both positive -> finishes there, 4 steps emitted
    assert steps == 4 and mel.shape == (B, 4, M * r) and stop.shape == (B, 4, 1)
    script2 = [[pos, neg]] * 8 + [[pos, pos]] * 5
    (mel, _), _, steps = O.dynamic_decode(_cell_with_stop_script(script2, B, M, r),
                                          O.StopTokenBasedInferenceHelper(B, M, r, 1, min_iters=2), 0, 50)
    assert steps == 9                       # one row alone never stops the batch
    (mel, _), _, steps = O.dynamic_decode(_cell_with_stop_script([[neg, neg]] * 50, B, M, r),
                                          O.StopTokenBasedInferenceHelper(B, M, r, 1, min_iters=2), 0, 7)
    assert steps == 7                       # maximum_iterations cap (tacotron_v1.py:150)
    # logit exactly 0 -> sigmoid = 0.5, not > 0.5
    (mel, _), _, steps = O.dynamic_decode(_cell_with_stop_script([[0.0, 0.0]] * 50, B, M, r),
                                          O.StopTokenBasedInferenceHelper(B, M, r, 1, min_iters=2), 0, 6)
    assert steps == 6


def test_feed_back_slice_is_last_nf_frames():
    """helpers.py:86,182: next input = last n_feed_frame frames of the r-group."""
    B, M, r, nf = 2, 3, 3, 2
    tgt = np.arange(B * 6 * M, dtype=np.float32).reshape(B, 6, M)
    h = O.TrainingHelper(tgt, M, r, nf)
    fin, first = h.initialize()
    assert first.shape == (B, M * nf) and not first.any() and not fin.any()
    fin, nxt, _ = h.next_inputs(0, (None, None), None)
    np.testing.assert_array_equal(nxt, tgt[:, 1:3].reshape(B, -1))
    assert not fin.any()
    fin, nxt, _ = h.next_inputs(1, (None, None), None)
    assert fin.all()                        # time + 1 >= S for every row, padding not masked
    with pytest.raises(AssertionError):
        O.TrainingHelper(tgt, M, 2, 3)


def test_inference_helper_zero_row():
    h = O.InferenceHelper(2, 3, 2, 1)
    out = np.zeros((2, 6), np.float32)
    out[1, 2] = 1.0
    fin, nxt, _ = h.next_inputs(0, (out, None), None)
    assert fin.tolist() == [True, False] and nxt.shape == (2, 3)


def test_full_forward_shapes_and_masking():
    d = PROFILE_TINY
    w = init_weights(d, 11, perturb=True)
    r = np.random.default_rng(0)
    ids = r.integers(0, d.num_symbols, (3, 7))
    lens = np.array([7, 5, 3])
    tgt = r.uniform(0, 1, (3, 8, d.num_mels)).astype(np.float32)
    out = O.tacotron2_forward(d, w, ids, lens, mode="eval", target=tgt, teacher_forcing=True)
    S = 8 // d.outputs_per_step
    assert out["mel"].shape == (3, 8, d.num_mels) and out["stop"].shape == (3, S, 1)
    assert out["alignment"].shape == (3, 7, S) and out["mel_postnet"].shape == out["mel"].shape
    assert np.all(out["alignment"][1, 5:] == 0) and np.all(out["memory"][2, 3:] == 0)
    # conv stack ignores lengths (tacotron_v2.py:54): changing a padded id changes row 1's memory
    ids2 = ids.copy()
    ids2[1, 5] = (ids2[1, 5] + 1) % d.num_symbols
    out2 = O.tacotron2_forward(d, w, ids2, lens, mode="eval", target=tgt, teacher_forcing=True)
    assert np.abs(out2["memory"][1, :5] - out["memory"][1, :5]).max() > 0
