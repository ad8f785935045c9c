"""Kernels whose workgroups wait for each other (csrc/persist.h, persist_bwd.h) bound every wait (250 ms) and raise an error
word when one runs out -- which happens when their workgroups are not all resident, e.g. another process holds part of the
GPU.  Contract (include/t2mi.h "Persistent kernels"): the call that saw the word re-runs its work on the launch-per-phase
schedule and returns VALID results; the event is counted (stat persist_timeouts) and the model's next 16 decodes skip the
persistent kernels.  Within one process such launches are serialised per GPU on the device, whatever streams they are on.
The tests force the word with option inject_timeout (it marks the next N persistent launches after they were enqueued)."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import t2_oracle as O
from tests.paper_cases import paper_case, train_case


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def setup():
    from tacotron2_amd.engine import Engine
    d, w, ids, lens = paper_case(91, 65, sharp=True, B=8)
    eng = Engine(d, w)
    yield d, w, eng, ids, lens
    eng.close()


def test_timed_out_free_decode_is_rerun_launch_per_phase(setup):
    d, w, eng, ids, lens = setup
    S = 14
    eng.set_option("persist", 1)
    memory = eng.encode(ids, lens)
    mel0, stop0, align0, n0 = eng.decode_free(memory, lens, max_iters=S, min_iters=100)
    mel0, stop0, align0 = _np(mel0).copy(), _np(stop0).copy(), _np(align0).copy()
    t0, p0 = eng.get_stat("persist_timeouts"), eng.get_stat("persist_launches")
    eng.set_option("inject_timeout", 1)
    mel, stop, align, n = eng.decode_free(memory, lens, max_iters=S, min_iters=100)
    assert eng.get_stat("persist_timeouts") == t0 + 1 and eng.get_stat("persist_launches") == p0 + 1
    assert n == n0 == S
    np.testing.assert_allclose(_np(mel), mel0, atol=1e-4)
    np.testing.assert_allclose(_np(stop), stop0, atol=1e-4)
    np.testing.assert_allclose(_np(align), align0, atol=1e-4)
    # hold-off: the next decodes take the launch schedule without trying; re-enabling the option clears it
    eng.decode_free(memory, lens, max_iters=S, min_iters=100)
    assert eng.get_stat("persist_launches") == p0 + 1
    eng.set_option("persist", 1)
    # split entry points: launch + finish
    bufs = eng.decode_free_buffers(len(lens), ids.shape[1], S)
    eng.set_option("inject_timeout", 1)
    eng.decode_free_launch(memory, lens, S, 100, bufs)
    assert eng.decode_free_finish() == S
    assert eng.get_stat("persist_timeouts") == t0 + 2 and eng.get_stat("persist_launches") == p0 + 2
    np.testing.assert_allclose(_np(bufs[0]), mel0, atol=1e-4)
    np.testing.assert_allclose(_np(bufs[2]), align0, atol=1e-4)
    eng.set_option("persist", 1)


def test_timed_out_teacher_decode_and_encoder_are_rerun(setup):
    d, w, eng, ids, lens = setup
    S = 6
    eng.set_option("persist", 1)
    target = np.random.default_rng(3).uniform(0, 1, (len(lens), S * d.outputs_per_step, d.num_mels)).astype(np.float32)
    memory_o = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    mel_o, stop_o, st = O.decoder_v2(d, w, memory_o, is_training=True, memory_sequence_length=lens, target=target)
    t0 = eng.get_stat("persist_timeouts")
    eng.set_option("inject_timeout", 1)
    memory = eng.encode(ids, lens)                                   # persistent BiLSTM marked as timed out -> launch per step
    assert eng.get_stat("persist_timeouts") == t0 + 1
    np.testing.assert_allclose(_np(memory), memory_o, atol=2e-4)
    eng.set_option("persist", 1)
    for tf in (True, False):
        p0 = eng.get_stat("persist_launches")
        eng.set_option("inject_timeout", 1)
        mel, stop, align = eng.decode_teacher(memory, lens, target, teacher_forcing=tf)
        assert eng.get_stat("persist_launches") == p0 + 1 and eng.get_stat("persist_timeouts") == t0 + (2 if tf else 3)
        eng.set_option("persist", 1)
        mel2, stop2, align2 = eng.decode_teacher(memory, lens, target, teacher_forcing=tf)
        assert eng.get_stat("persist_launches") == p0 + 2
        np.testing.assert_allclose(_np(mel), _np(mel2), atol=1e-4)
        np.testing.assert_allclose(_np(align), _np(align2), atol=1e-4)
        if tf:
            np.testing.assert_allclose(_np(mel), mel_o, atol=2e-4)
            np.testing.assert_allclose(_np(stop), stop_o, atol=2e-4)
            np.testing.assert_allclose(_np(align), np.stack(st[0].alignment_history, 0), atol=2e-4)


def test_timed_out_training_recurrences_are_rerun():
    """Forward decoder, backward decoder, encoder forward and encoder backward of the training step, each marked as timed out in
    turn: the gradients equal those of the undisturbed step."""
    from tacotron2_amd import training as TR
    from tacotron2_amd.engine import Engine
    B, T, S = 6, 40, 3
    d, w, ids, lens, labels = train_case(B, T, S)
    eng = Engine(d, w)
    tr = TR.Trainer(eng, B, T, S)
    try:
        spec = TR.flatten_spec(w)
        tr.bind_gradients(spec)

        def step(inject_at):
            flat = torch.zeros(TR.flat_size(spec), dtype=torch.float32, device="cuda")
            calls = [lambda: tr.encoder_forward(ids, lens, drop_rate=0.5, seed=3),
                     lambda: tr.decoder_forward(step.memory, lens, labels["mel"], prenet_drop_rate=0.5, seed=3),
                     lambda: tr.decoder_backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], flat),
                     lambda: tr.encoder_backward(step.dmem, flat)]
            for i, call in enumerate(calls):
                eng.set_option("persist", 1)                       # clears the hold-off of an earlier injection
                if i == inject_at:
                    eng.set_option("inject_timeout", 1)
                out = call()
                if i == 0:
                    step.memory = out[0]
                if i == 2:
                    step.dmem, losses = out
            return TR.unflatten(flat, spec), float(losses[2])

        g0, l0 = step(-1)
        assert eng.get_stat("persist_timeouts") == 0
        for k in range(4):
            g1, l1 = step(k)
            assert eng.get_stat("persist_timeouts") == k + 1, k
            assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
            for name in sorted(g0):
                if "moving_" in name:
                    continue
                scale = float(np.abs(g0[name]).max())
                assert float(np.abs(g1[name] - g0[name]).max()) <= 2e-4 * scale + 1e-9, (k, name)
    finally:
        tr.close()
        eng.close()


def test_two_engines_on_two_streams_decode_concurrently():
    """Two engines (two library streams), driven from two host threads at the same time: their persistent launches are
    serialised on the device by the library (XwgGate), both return the oracle's results and nothing times out."""
    from tacotron2_amd.engine import Engine
    S = 10
    cases = [paper_case(91, 65, sharp=True, B=64), paper_case(77, 64, sharp=True, B=20, full_row=5)]
    engines = [Engine(c[0], c[1]) for c in cases]
    refs = []
    for d, w, ids, lens in cases:
        mem = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
        mel_o, stop_o, st = O.decoder_v2(d.replace(min_iters=100), w, mem, memory_sequence_length=lens, max_iters=S)
        refs.append((mem, mel_o, np.stack(st[0].alignment_history, 0)))
    results, errors = [None, None], []
    barrier = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng, (d, w, ids, lens) = engines[i], cases[i]
                barrier.wait()
                for _ in range(6):                                  # several rounds so that the launches really interleave
                    memory = eng.encode(ids, lens)
                    mel, stop, align, n = eng.decode_free(memory, lens, max_iters=S, min_iters=100)
                torch.cuda.current_stream().synchronize()
                results[i] = (_np(memory), _np(mel), _np(align), n)
        except Exception as ex:                                     # noqa: BLE001
            errors.append((i, repr(ex)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for i in range(2):
        mem, mel, align, n = results[i]
        assert n == S
        np.testing.assert_allclose(mem, refs[i][0], atol=2e-4)
        np.testing.assert_allclose(mel, refs[i][1], atol=2e-4)
        np.testing.assert_allclose(align, refs[i][2], atol=2e-4)
        assert engines[i].get_stat("persist_timeouts") == 0
        assert engines[i].get_stat("persist_launches") == 6
    for e in engines:
        e.close()
