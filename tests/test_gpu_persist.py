"""The persistent decoder (csrc/persist.h: the whole free-running dynamic_decode loop as ONE launch, paper-size
dimension class) against the oracle and against the launch-per-phase schedule it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import t2_oracle as O
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
from tests.golden import make_golden as G
from tests.paper_cases import paper_case


@pytest.fixture(scope="module", params=["sharp", "init"])
def paper(request):
    """'sharp': peaked, moving attention (weights.sharpen_attention) -- the setting in which every term of the attention score
    is visible at the tolerances below (tests/test_oracle_sensitivity.py proves that on the oracle for these very inputs);
    'init': TF's fresh-graph initialisation (near-uniform attention), kept as the second weight regime."""
    from tacotron2_amd.engine import Engine
    d, w, ids, lens = paper_case(91, 65, sharp=(request.param == "sharp"))
    eng = Engine(d, w)
    memory_o = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    # HIP-vs-HIP comparisons (two schedules of the same fp32 arithmetic, different summation orders): with a peaked attention
    # the recurrence amplifies rounding differences to ~1e-5 within tens of steps (the oracle's own fp32-vs-fp64 difference
    # on these weights, weights.sharpen_attention), with the near-uniform one they stay ~1e-7
    eng.twin_atol = (1e-4, 1e-4) if request.param == "sharp" else (1e-4, 1e-5)          # (mel / stop, alignments)
    yield d, w, eng, memory_o, lens
    eng.close()


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("B", [1, 5, 16, 20, 33, 64])
def test_persistent_free_run_against_oracle(paper, B):
    """1, 2 and 4 row blocks, every attention split (16, 8, 4 workgroups per row), ragged lengths."""
    d, w, eng, memory_o, lens = paper
    eng.set_option("persist", 1)
    S = 7
    T = int(lens[:B].max())
    mem = memory_o[:B, :T]
    mel_o, stop_o, st = O.decoder_v2(d.replace(min_iters=100), w, mem, memory_sequence_length=lens[:B], max_iters=S)
    before = eng.get_stat("persist_launches")
    mel, stop, align, n = eng.decode_free(mem, lens[:B], max_iters=S, min_iters=100)
    assert eng.get_stat("persist_launches") == before + 1, "the persistent kernel did not run"
    assert n == S
    np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)
    np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
    np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)


@pytest.mark.parametrize("B", [1, 64])
def test_persistent_equals_launch_schedule(paper, B):
    """Same decode, 48 free-running steps, one persistent launch against six launches per step."""
    d, w, eng, memory_o, lens = paper
    S = 48
    T = int(lens[:B].max())
    mem = memory_o[:B, :T]
    eng.set_option("persist", 0)
    mel0, stop0, align0, n0 = eng.decode_free(mem, lens[:B], max_iters=S, min_iters=100)
    mel0, stop0, align0 = _np(mel0).copy(), _np(stop0).copy(), _np(align0).copy()
    eng.set_option("persist", 1)
    mel1, stop1, align1, n1 = eng.decode_free(mem, lens[:B], max_iters=S, min_iters=100)
    assert n0 == n1 == S
    np.testing.assert_allclose(_np(mel1), mel0, atol=eng.twin_atol[0])
    np.testing.assert_allclose(_np(stop1), stop0, atol=eng.twin_atol[0])
    np.testing.assert_allclose(_np(align1), align0, atol=eng.twin_atol[1])


@pytest.mark.parametrize("B", [1, 64])
def test_persistent_stop_index_at_paper_sizes(paper, B):
    """SURVEY.md 8c fixture (4) at paper sizes: the batch-coupled stop rule (helpers.py:90-94) fires mid-sequence; the step
    count must equal the oracle's, with every row's decision clear of the tolerance around the crossing."""
    from tacotron2_amd.engine import Engine
    d, w, _, memory_o, lens = paper
    T = int(lens[:B].max())
    mem = memory_o[:B, :T]
    # the random-init stop logit drifts by ~5e-3 per step: scale the stop projection so that consecutive decisions differ by
    # ~0.1, then put the threshold into the widest gap of the per-step row-minimum (tests/golden/make_golden.py:tune_stop)
    w = dict(w)
    w["decoder/stop_token_projection/kernel"] = (w["decoder/stop_token_projection/kernel"] * 25.0).astype(np.float32)
    w2, steps_o = G.tune_stop(d, w, mem, lens[:B], 13, 24, min_gap=2.5e-2)
    mel_r, stop_r, _ = O.decoder_v2(d, w2, mem, memory_sequence_length=lens[:B], max_iters=24)
    assert steps_o == mel_r.shape[1] // d.outputs_per_step
    k = steps_o - 1
    m2 = stop_r[..., 0].min(axis=0)
    assert float(m2[k]) > 1e-2 and float(m2[11:k].max()) < -1e-2, "fixture too close to the decision boundary"
    eng2 = Engine(d, w2)
    try:
        for persist in (1, 0):
            eng2.set_option("persist", persist)
            mel, stop, align, n = eng2.decode_free(mem, lens[:B], max_iters=24)
            assert n == steps_o, (persist, n, steps_o)
            np.testing.assert_allclose(_np(mel), mel_r, atol=1e-3)
            np.testing.assert_allclose(_np(stop), stop_r, atol=1e-3)
    finally:
        eng2.close()


def test_persistent_validation_free_run(paper):
    """ValidationHelper(teacher_forcing=False) (helpers.py:131-138): fixed step count, own output fed back."""
    d, w, eng, memory_o, lens = paper
    B, S = 9, 6
    T = int(lens[:B].max())
    mem = memory_o[:B, :T]
    target = np.zeros((B, S * d.outputs_per_step, d.num_mels), np.float32)
    eng.set_option("persist", 1)
    mel, stop, align = eng.decode_teacher(mem, lens[:B], target, teacher_forcing=False)
    mel_o, stop_o, st = O.decoder_v2(d, w, mem, is_validation=True, teacher_forcing=False, memory_sequence_length=lens[:B],
                                     target=target, max_iters=S)
    np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
    np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)
    np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)


@pytest.mark.parametrize("B", [1, 20, 64])
def test_persistent_teacher_forced_against_oracle_and_launch_schedule(paper, B):
    """TrainingHelper / ValidationHelper(teacher_forcing=True) (helpers.py:131-136, 176-184) as one launch: PreNets of all steps
    before it, projections after it; against the oracle, the launch-per-phase schedule, and with the final state."""
    d, w, eng, memory_o, lens = paper
    S = 9
    T = int(lens[:B].max())
    mem = memory_o[:B, :T]
    target = np.random.default_rng(3).uniform(-1, 1, (B, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    mel_o, stop_o, st = O.decoder_v2(d, w, mem, is_training=True, memory_sequence_length=lens[:B], target=target)
    eng.set_option("persist", 1)
    before = eng.get_stat("persist_launches")
    mel, stop, align = eng.decode_teacher(mem, lens[:B], target, teacher_forcing=True)
    assert eng.get_stat("persist_launches") == before + 1, "the persistent kernel did not run"
    np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)
    np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
    np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)
    for name, want in (("attention_c", st[0].cell_state[0]), ("attention_h", st[0].cell_state[1]), ("attention", st[0].attention),
                       ("attention_state", st[0].attention_state), ("lstm1_c", st[1][0]), ("lstm2_h", st[2][1])):
        np.testing.assert_allclose(_np(eng.decode_final_state(name, B, T)), want, atol=2e-4, err_msg=name)
    mel1, stop1, align1 = _np(mel).copy(), _np(stop).copy(), _np(align).copy()
    eng.set_option("persist", 0)
    mel0, stop0, align0 = eng.decode_teacher(mem, lens[:B], target, teacher_forcing=True)
    eng.set_option("persist", 1)
    np.testing.assert_allclose(mel1, _np(mel0), atol=eng.twin_atol[0])
    np.testing.assert_allclose(stop1, _np(stop0), atol=eng.twin_atol[0])
    np.testing.assert_allclose(align1, _np(align0), atol=eng.twin_atol[1])
    # the two-chain schedule (two hops per step, csrc/persist.h MODE 3: the default) against the four-hop schedule (option tf2 = 0),
    # and an even step count: the last step's hidden states then live in the second bank and are republished for final_state
    for steps in (S, S - 1):
        tgt = target[:, :steps * d.outputs_per_step]
        got = {}
        for tf2 in (1, 0):
            eng.set_option("tf2", tf2)
            before = eng.get_stat("persist_launches")
            out = eng.decode_teacher(mem, lens[:B], tgt, teacher_forcing=True)
            assert eng.get_stat("persist_launches") == before + 1 and eng.get_stat("persist_timeouts") == 0
            got[tf2] = [_np(o).copy() for o in out] + [_np(eng.decode_final_state(n, B, T)).copy()
                                                       for n in ("attention_h", "attention_output", "lstm1_h", "lstm2_h", "lstm2_c", "attention")]
        eng.set_option("tf2", 1)
        for a1, a0 in zip(got[1], got[0]):                                   # another summation order of the hoisted K ranges
            np.testing.assert_allclose(a1, a0, atol=max(eng.twin_atol))


def test_persistent_decode_after_a_device_side_parameter_refresh():
    """t2_trainer_load_params rebuilds every kernel-native operand on the device, the composed projection + first-PreNet columns of
    the free-running decode included (compose_proj_kernel): the refreshed engine still decodes as ONE launch and equals an engine
    built from the new parameters on the host."""
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.engine import Engine
    d = PROFILE_P
    w = init_weights(d, seed=17, perturb=True)
    eng = Engine(d, w)
    tr = TR.Trainer(eng, 2, 12, 2)
    spec = TR.flatten_spec(w)
    tr.bind_gradients(spec)
    rng = np.random.default_rng(4)
    w2 = {}
    for k, v in w.items():
        moved = not k.endswith("moving_variance")
        w2[k] = (v + (0.02 * rng.standard_normal(v.shape).astype(np.float32) if moved else 0.0)).astype(np.float32)
    tr.load_params(torch.tensor(TR.flatten(w2, spec), device="cuda"))
    B, T, S = 3, 30, 12
    lens = np.array([30, 17, 25])
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    before = eng.get_stat("persist_launches")
    mel, stop, align, n = eng.decode_free(memory, lens, max_iters=S, min_iters=100)
    assert eng.get_stat("persist_launches") == before + 1, "decode fell back to the launch schedule after the refresh"
    eng2 = Engine(d, w2)
    mel2, stop2, align2, n2 = eng2.decode_free(memory, lens, max_iters=S, min_iters=100)
    assert n == n2 == S
    np.testing.assert_allclose(_np(mel), _np(mel2), atol=1e-5)
    np.testing.assert_allclose(_np(stop), _np(stop2), atol=1e-5)
    np.testing.assert_allclose(_np(align), _np(align2), atol=1e-6)
    mel_o, stop_o, _ = O.decoder_v2(d.replace(min_iters=100), w2, memory, memory_sequence_length=lens, max_iters=S)
    np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
    for o in (tr, eng2, eng):
        o.close()


@pytest.mark.parametrize("B", [1, 20, 64])
def test_persistent_encoder_bilstm_against_oracle_and_launch_schedule(B):
    """EncoderV2 (tacotron_v2.py:53-63) with the bidirectional ZoneoutLSTM as ONE launch (persist.h encoder_kernel: weights in
    registers, h exchanged in-launch) against the oracle and against the launch-per-step schedule; ragged lengths, a row of
    length 1 and a full-length row."""
    from tacotron2_amd.engine import Engine
    d = PROFILE_P
    w = init_weights(d, seed=23, perturb=True)
    eng = Engine(d, w)
    try:
        rng = np.random.default_rng(B)
        T = 37
        lens = rng.integers(5, T + 1, size=B).astype(np.int64)
        lens[0] = T
        if B > 2:
            lens[1] = 1
        ids = np.zeros((B, T), np.int64)
        for b, n in enumerate(lens):
            ids[b, :n] = rng.integers(1, d.num_symbols, size=n)
        want = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
        eng.set_option("enc_persist", 1)
        before = eng.get_stat("enc_persist_launches")
        got = _np(eng.encode(ids, lens)).copy()
        assert eng.get_stat("enc_persist_launches") == before + 1, "the persistent encoder kernel did not run"
        np.testing.assert_allclose(got, want, atol=2e-4)
        eng.set_option("enc_persist", 0)
        ref = _np(eng.encode(ids, lens))
        assert eng.get_stat("enc_persist_launches") == before + 1
        np.testing.assert_allclose(got, ref, atol=1e-5)
        for b, n in enumerate(lens):                       # dynamic_rnn: zero output past the row's length
            assert not got[b, n:].any()
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,S", [(1, 9, 3), (3, 2, 1), (20, 77, 7), (16, 200, 3), (64, 140, 5)])
def test_persistent_backward_matches_launch_per_phase(B, T, S):
    """The training backward's reverse-time recurrences as persistent launches (csrc/persist_bwd.h: decoder cell stack +
    attention, and the encoder BiLSTM; option bwd_persist) and the encoder's training forward on pz::encoder_kernel
    (option enc_persist) against the launch-per-phase schedule on the same inputs and masks: every gradient and d memory
    to 2e-4 of its scale (the launch-per-phase backward is checked against the fp64 autograd oracle in test_gpu_train.py /
    test_gpu_fullsize.py; at B = 32 the persistent one is too, there)."""
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.engine import Engine
    d = PROFILE_P
    w = init_weights(d, seed=5, perturb=True)
    rng = np.random.default_rng(B * 1000 + T + S)
    r = d.outputs_per_step
    lens = rng.integers(max(1, T // 2), T + 1, B).astype(np.int64)
    lens[0] = T
    ids = rng.integers(0, d.num_symbols, (B, T))
    tl = rng.integers(max(1, S * r // 2), S * r + 1, B)
    tl[0] = S * r
    sl = (tl + r - 1) // r
    mel = rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32)
    done = (np.arange(S)[None, :] >= (sl[:, None] - 1)).astype(np.float32)
    smask = (np.arange(S * r)[None, :] < tl[:, None]).astype(np.float32)
    bmask = (np.arange(S)[None, :] < sl[:, None]).astype(np.float32)
    eng = Engine(d, w)
    tr = TR.Trainer(eng, B, T, S)
    try:
        spec = TR.flatten_spec(w)
        tr.bind_gradients(spec)
        out = {}
        for opt in (0, 1):
            eng.set_option("bwd_persist", opt)
            eng.set_option("enc_persist", opt)
            n_b, n_e = eng.get_stat("bwd_persist_launches"), eng.get_stat("enc_persist_launches")
            memory, _ = tr.encoder_forward(ids, lens, drop_rate=0.5, seed=3)
            tr.decoder_forward(memory, lens, mel, prenet_drop_rate=0.5, seed=3)
            flat = torch.zeros(TR.flat_size(spec), dtype=torch.float32, device="cuda")
            dmem, losses = tr.decoder_backward(done, smask, bmask, flat)
            tr.encoder_backward(dmem, flat)
            assert eng.get_stat("bwd_persist_launches") == n_b + 2 * opt          # decoder + encoder recurrences
            assert eng.get_stat("enc_persist_launches") == n_e + opt
            out[opt] = (TR.unflatten(flat, spec), _np(dmem), _np(memory), float(losses[2]))
            if opt == 1:
                # fixed ownership and summation orders, no float atomics: a second run of the same step is bit-identical
                memory2, _ = tr.encoder_forward(ids, lens, drop_rate=0.5, seed=3)
                tr.decoder_forward(memory2, lens, mel, prenet_drop_rate=0.5, seed=3)
                flat2 = torch.zeros_like(flat)
                dmem2, _ = tr.decoder_backward(done, smask, bmask, flat2)
                tr.encoder_backward(dmem2, flat2)
                assert torch.equal(flat2, flat) and torch.equal(dmem2, dmem)
        g0, m0, mem0, l0 = out[0]
        g1, m1, mem1, l1 = out[1]
        np.testing.assert_allclose(mem1, mem0, atol=1e-5)
        assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
        np.testing.assert_allclose(m1, m0, atol=2e-4 * float(np.abs(m0).max()) + 1e-9)
        for name in sorted(g0):
            if "moving_" in name:
                continue
            scale = float(np.abs(g0[name]).max())
            assert float(np.abs(g1[name] - g0[name]).max()) <= 2e-4 * scale + 1e-9, name
    finally:
        tr.close()
        eng.close()


@pytest.mark.gpu
def test_persistent_training_recurrences_after_parameter_refresh():
    """t2_trainer_load_params rebuilds the persistent backward kernels' operands (2-D W^T tiles of the decoder LSTMs, the query
    tile, the encoder's column-owner tiles) from the flat parameter buffer on the device: a trainer refreshed with new
    parameters must produce the gradients of a trainer created from them."""
    import torch
    from tacotron2_amd import training as TR
    from tacotron2_amd.engine import Engine
    d = PROFILE_P
    w_old = init_weights(d, seed=5, perturb=True)
    w_new = init_weights(d, seed=77, perturb=True)
    B, T, S, r = 20, 61, 4, d.outputs_per_step
    rng = np.random.default_rng(2)
    lens = rng.integers(30, T + 1, B).astype(np.int64)
    lens[0] = T
    ids = rng.integers(0, d.num_symbols, (B, T))
    mel = rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32)
    done = np.zeros((B, S), np.float32)
    done[:, -1] = 1
    smask, bmask = np.ones((B, S * r), np.float32), np.ones((B, S), np.float32)
    spec = TR.flatten_spec(w_old)

    def step(eng, tr):
        n0 = eng.get_stat("bwd_persist_launches")
        memory, _ = tr.encoder_forward(ids, lens, drop_rate=0.5, seed=9)
        tr.decoder_forward(memory, lens, mel, prenet_drop_rate=0.5, seed=9)
        flat = torch.zeros(TR.flat_size(spec), dtype=torch.float32, device="cuda")
        dmem, losses = tr.decoder_backward(done, smask, bmask, flat)
        tr.encoder_backward(dmem, flat)
        assert eng.get_stat("bwd_persist_launches") == n0 + 2
        return TR.unflatten(flat, spec), float(losses[2])

    eng_a = Engine(d, w_old)
    tr_a = TR.Trainer(eng_a, B, T, S)
    eng_b = Engine(d, w_new)
    tr_b = TR.Trainer(eng_b, B, T, S)
    try:
        tr_a.bind_gradients(spec)
        tr_b.bind_gradients(spec)
        step(eng_a, tr_a)                                                     # builds the operands from the old weights
        tr_a.load_params(torch.from_numpy(TR.flatten(w_new, spec)).cuda())    # ... and replaces them on the device
        g_a, loss_a = step(eng_a, tr_a)
        g_b, loss_b = step(eng_b, tr_b)
        assert abs(loss_a - loss_b) <= 1e-5 * max(1.0, abs(loss_b))
        for name in sorted(g_b):
            if "moving_" in name:
                continue
            scale = float(np.abs(g_b[name]).max())
            assert float(np.abs(g_a[name] - g_b[name]).max()) <= 2e-4 * scale + 1e-9, name
    finally:
        tr_a.close(); tr_b.close()
        eng_a.close(); eng_b.close()


@pytest.mark.parametrize("B,T,eligible", [(1, 9, True), (3, 2, True), (20, 77, True), (1, 590, True), (64, 256, True), (64, 260, False),
                                          (16, 257, True), (1, 1000, True), (2, 2000, False), (40, 700, False), (17, 3001, False)])
def test_persistent_forward_at_edge_shapes_against_oracle(B, T, eligible):
    """Forward parity where the attention split has corner cases: chunks with fewer than 16 or zero positions
    (T < 16 C), T not a multiple of 4, the longest eligible inputs (ceil(T / C) = 64 and the LDS plan's limit), and just past
    the eligibility edge, where the decode must take the launch-per-phase schedule (persist_launches does not advance).
    Long inputs (round 3; the reference has no length limit, tacotron_v2.py:118-141): T_in of 700 ... 3001 on the launch-per-phase
    schedule, whose attention kernel sweeps the location features in chunks of 256 positions (whole and ragged last chunks,
    one and several row blocks), and T_in = 1000 on the persistent kernel.
    Free-running and teacher-forced, against the oracle; peaked attention."""
    from tacotron2_amd.engine import Engine
    from tacotron2_amd.weights import sharpen_attention
    d = PROFILE_P
    w = sharpen_attention(init_weights(d, seed=31, perturb=True))
    rng = np.random.default_rng(B * 1000 + T)
    lens = rng.integers(max(1, T // 2), T + 1, size=B).astype(np.int64)
    lens[0] = T
    if B > 2:
        lens[1] = 1                                                   # a row with a single position
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    S = 4
    eng = Engine(d, w)
    try:
        mel_o, stop_o, st = O.decoder_v2(d.replace(min_iters=100), w, memory, memory_sequence_length=lens, max_iters=S)
        n0 = eng.get_stat("persist_launches")
        mel, stop, align, n = eng.decode_free(memory, lens, max_iters=S, min_iters=100)
        assert eng.get_stat("persist_launches") == n0 + (1 if eligible else 0)
        assert n == S
        np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)
        np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
        np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)
        target = rng.uniform(0, 1, (B, S * d.outputs_per_step, d.num_mels)).astype(np.float32)
        mel_o, stop_o, st = O.decoder_v2(d, w, memory, is_training=True, memory_sequence_length=lens, target=target)
        mel, stop, align = eng.decode_teacher(memory, lens, target, teacher_forcing=True)
        assert eng.get_stat("persist_launches") == n0 + (2 if eligible else 0)
        np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)
        np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
        np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)
        for b in range(B):                                            # masked positions get exactly zero weight
            assert not _np(align)[:, b, lens[b]:].any()
    finally:
        eng.close()
