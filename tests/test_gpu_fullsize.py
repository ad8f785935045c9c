"""Parity at BASELINE.json's full sizes (profile P, B = 64 per GPU, ragged T_in in 60..140) and the edge cases of
the path: directly against the oracle for as many decoder steps as the oracle finishes in seconds, and through
size-independent properties (batch-row independence, the reference's free-run -> teacher-forcing property)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import t2_oracle as O
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
from tests.paper_cases import paper_case, train_case

ATOL = 1e-3


@pytest.fixture(scope="module")
def paper():
    """Peaked, moving attention (weights.sharpen_attention): tests/test_oracle_sensitivity.py shows on the oracle that every
    term of the attention score moves these alignments by >= 10 x the tolerances used below."""
    from tacotron2_amd.engine import Engine
    d, w, ids, lens = paper_case(77, 64, sharp=True, full_row=5)
    eng = Engine(d, w)
    yield d, w, eng, ids, lens
    eng.close()


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("hoist", [1, 0])
def test_full_size_batch64_against_oracle(paper, hoist):
    d, w, eng, ids, lens = paper
    eng.set_option("hoist", hoist)
    S = 8
    rng = np.random.default_rng(3)
    target = rng.uniform(0, 1, (64, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    memory_o = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    memory = eng.encode(ids, lens)
    np.testing.assert_allclose(_np(memory), memory_o, atol=ATOL)
    mel_o, stop_o, st = O.decoder_v2(d, w, memory_o, is_validation=True, teacher_forcing=True, memory_sequence_length=lens,
                                     target=target, max_iters=S)
    mel, stop, align = eng.decode_teacher(memory, lens, target, teacher_forcing=True)
    np.testing.assert_allclose(_np(mel), mel_o, atol=ATOL)
    np.testing.assert_allclose(_np(stop), stop_o, atol=ATOL)
    np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=ATOL)
    eng.set_option("hoist", 1)


@pytest.mark.parametrize("spread,projfuse", [(2, 1), (2, 0), (1, 1), (0, 1), (0, 0)])
@pytest.mark.parametrize("B", [64, 24, 3])
def test_full_size_free_run_schedules_against_oracle(paper, spread, projfuse, B):
    """Free-running decode under every launch schedule (riders spread over the small launches or not; first PreNet layer
    composed into the projection launch or not) against the oracle's free run, for 1, 2 and 4 row blocks."""
    d, w, eng, ids, lens = paper
    eng.set_option("spread", spread)
    eng.set_option("projfuse", projfuse)
    S = 6
    memory_o = O.encoder_v2(d, w, O.embedding_lookup(ids[:B], w["embedding"]), lens[:B])
    mel_o, stop_o, st = O.decoder_v2(d.replace(min_iters=100), w, memory_o, memory_sequence_length=lens[:B], max_iters=S)
    mel, stop, align, n = eng.decode_free(memory_o, lens[:B], max_iters=S, min_iters=100)
    assert n == S
    np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
    np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)
    np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)
    eng.set_option("spread", 2)
    eng.set_option("projfuse", 1)


def test_full_size_rows_are_independent(paper):
    """No tensor is shared across batch rows: rows decoded alone equal the same rows of the B = 64 batch."""
    d, w, eng, ids, lens = paper
    S = 24
    rng = np.random.default_rng(4)
    target = rng.uniform(0, 1, (64, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    memory = eng.encode(ids, lens)
    mel, stop, align = eng.decode_teacher(memory, lens, target, teacher_forcing=True)
    for b in (0, 17, 63):
        T_b = int(lens[b])
        # same padded width: the conv stack ignores lengths, so padding leaks into the last real positions
        # (modules.py:149-157) and a row re-encoded at its own length would differ there
        mem_b = eng.encode(ids[b:b + 1], lens[b:b + 1])
        np.testing.assert_allclose(_np(mem_b)[0], _np(memory)[b], atol=1e-5)
        mel_b, stop_b, align_b = eng.decode_teacher(mem_b, lens[b:b + 1], target[b:b + 1], teacher_forcing=True)
        np.testing.assert_allclose(_np(mel_b)[0], _np(mel)[b], atol=1e-4)
        np.testing.assert_allclose(_np(stop_b)[0], _np(stop)[b], atol=1e-4)
        np.testing.assert_allclose(_np(align_b)[:, 0, :], _np(align)[:, b, :], atol=1e-4)    # peaked attention: see test_gpu_persist.paper
        assert float(np.abs(_np(align)[:, b, T_b:]).max(initial=0.0)) == 0.0          # masked positions get exactly zero weight


def test_full_size_free_run_equals_teacher_forced(paper):
    """tests/decoder_test.py:42-67 at paper sizes, B = 64: the free-running output fed back as the target."""
    d, w, eng, ids, lens = paper
    memory = eng.encode(ids, lens)
    mel_f, stop_f, align_f, n = eng.decode_free(memory, lens, max_iters=20, min_iters=100)      # stop rule cannot fire
    assert n == 20
    mel_t, stop_t, align_t = eng.decode_teacher(memory, lens, mel_f, teacher_forcing=True)
    np.testing.assert_allclose(_np(mel_t), _np(mel_f), atol=1e-5)
    np.testing.assert_allclose(_np(stop_t), _np(stop_f), atol=1e-5)
    np.testing.assert_allclose(_np(align_t), _np(align_f), atol=1e-4)          # peaked attention: see test_gpu_persist.paper
    post = eng.postnet(mel_f[:8])
    np.testing.assert_allclose(_np(post), O.postnet_v2(d, w, _np(mel_f[:8])), atol=ATOL)


def test_edge_single_position_and_limits(paper):
    from tacotron2_amd._lib import T2Error
    d, w, eng, ids, lens = paper
    one = np.array([[40]], np.int64)
    mem = eng.encode(one, np.array([1]))
    mem_o = O.encoder_v2(d, w, O.embedding_lookup(one, w["embedding"]), np.array([1]))
    np.testing.assert_allclose(_np(mem), mem_o, atol=ATOL)
    mel, stop, align, n = eng.decode_free(mem, np.array([1]), max_iters=3, min_iters=100)
    assert n == 3 and np.allclose(_np(align), 1.0)                       # softmax over one position
    mel_o, _, _ = O.decoder_v2(d.replace(min_iters=100), w, mem_o, memory_sequence_length=np.array([1]), max_iters=3)
    np.testing.assert_allclose(_np(mel), mel_o, atol=ATOL)
    with pytest.raises(T2Error) as ei:                                   # longer than the attention kernel's LDS plan
        eng.decode_free(np.zeros((1, 7000, d.enc_units), np.float32), np.array([7000]), max_iters=2)    # ~6 400 since the chunked sweep
    assert ei.value.code == -1 and "too long" in ei.value.msg
    with pytest.raises(T2Error):                                         # length 0 is not a valid memory_sequence_length
        eng.decode_free(_np(mem), np.array([0]), max_iters=2)
    with pytest.raises(T2Error):                                         # T_out not a multiple of r
        eng.decode_teacher(mem, np.array([1]), np.zeros((1, 3, d.num_mels), np.float32))


def test_postnet_at_bench_shape_against_oracle(paper):
    """PostNetV2 (tacotron_v2.py:181-204) at the shape bench.py times: B = 64 x 800 frames at profile P, where launch_conv picks
    the 128-row tiles of conv_gemm_fast_kernel<2> (smaller inputs land on the 64-row variant)."""
    d, w, eng, ids, lens = paper
    rng = np.random.default_rng(11)
    mel = rng.uniform(-0.2, 1.2, (64, 800, d.num_mels)).astype(np.float32)
    post = eng.postnet(mel)
    np.testing.assert_allclose(_np(post), O.postnet_v2(d, w, mel), atol=2e-4)
    eng.set_option("conv_generic", 1)                    # and the generic conv kernel on the same input
    try:
        np.testing.assert_allclose(_np(eng.postnet(mel)), _np(post), atol=1e-5)
    finally:
        eng.set_option("conv_generic", 0)


def test_training_gradients_at_bench_shape_against_autograd():
    """The training step at the sizes bench.py times (profile P, B = 32, T_in = 140; 4 decoder steps keep the fp64 autograd
    reference to ~30 s): the profile-P instantiations of the backward kernels (bwd_dense_kernel<2> with static ring, nine
    position chunks per row in the attention backward at U = 128, split-K weight-gradient GEMMs) against torch.autograd.  The
    forward recurrence is the persistent decoder in teacher mode with the tape and sampled zoneout (csrc/persist.h)."""
    import torch
    from oracle import torch_train as TT
    from tacotron2_amd import training as TR
    from tacotron2_amd.engine import Engine
    B, T, S = 32, 140, 4
    d, w, ids, lens, labels = train_case(B, T, S)          # peaked attention: attention gradients >= 1e-5
    ref = TT.train_forward_backward(d, w, ids, lens, labels, seed=3, prenet_drop_rate=0.5, encoder_drop_rate=0.5)
    eng = Engine(d, w)
    tr = TR.Trainer(eng, B, T, S)
    spec = TR.flatten_spec(w)
    tr.bind_gradients(spec)
    memory, _ = tr.encoder_forward(ids, lens, drop_rate=0.5, seed=3)
    np.testing.assert_allclose(_np(memory), ref["memory"], atol=ATOL)
    before = eng.get_stat("persist_launches")
    mel, stop, align = tr.decoder_forward(memory, lens, labels["mel"], prenet_drop_rate=0.5, seed=3)
    assert eng.get_stat("persist_launches") == before + 1, "the training forward did not run as the persistent launch"
    np.testing.assert_allclose(_np(mel), ref["mel"], atol=ATOL)
    flat = torch.zeros(TR.flat_size(spec), dtype=torch.float32, device="cuda")
    dmem, losses = tr.decoder_backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], flat)
    tr.encoder_backward(dmem, flat)
    assert abs(float(losses[2]) - ref["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))
    grads = TR.unflatten(flat, spec)
    bad = []
    for name, g_ref in sorted(ref["grads"].items()):
        scale = float(np.abs(g_ref).max())
        err = float(np.abs(grads[name] - g_ref).max())
        if "moving_" not in name:
            assert scale >= 1e-6, (name, scale)          # the absolute floor below is <= 1e-3 of every tensor's scale
        if err > 2e-3 * scale + 1e-9:
            bad.append((name, err, scale))
    assert not bad, bad
    tr.close()
    eng.close()
