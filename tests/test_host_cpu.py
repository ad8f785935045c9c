"""CPU-side checks: the C-ABI library loads and exports every symbol include/t2mi.h declares, the
golden fixtures still match the oracle, host logic (helpers, sharding, constructor checks)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tacotron2_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "t2mi.h")).read()
    declared = set(re.findall(r"\b(t2(?:v1)?_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = _lib.load()                     # raises if a bound symbol is missing
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in t2mi.h but not exported"
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert lib.t2_version() == 1


def test_model_create_validation_without_gpu():
    """Argument validation happens before any device work: mirrors tacotron_v2.py:40, helpers.py:55,
    modules.py:241-245."""
    import ctypes as C
    from tacotron2_amd import _lib
    from tacotron2_amd.engine import _dims_struct
    from tacotron2_amd.hparams import PROFILE_TINY
    lib = _lib.load()
    out = C.c_void_p()
    assert lib.t2_model_create(None, None, C.byref(out)) == _lib.T2_EINVAL
    assert b"null" in lib.t2_last_error()
    ds = _dims_struct(PROFILE_TINY)
    assert ds.num_prenet_layers == 2 and ds.prenet_units[1] == 9


@pytest.mark.parametrize("name", ["tiny_r1_nf1", "tiny_r2_nf1", "tiny_r2_nf2", "mid_teacher", "mid_free"])
def test_oracle_matches_golden(name):
    from tests.golden import make_golden as G
    d, w, inp = G.materialize(name, G.cases()[name])
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    assert abs(G.weights_checksum(w) - float(gold["weights_checksum"])) < 1e-3 * float(gold["weights_checksum"])
    ref = G.run_oracle(d, w, inp)
    assert ref["steps"] == int(gold["steps"])
    for k in ("memory", "mel", "stop", "alignment", "mel_postnet"):
        np.testing.assert_allclose(ref[k], gold[k], atol=2e-5, err_msg=k)


def test_helpers_on_cpu_tensors_match_oracle_helpers():
    from oracle import t2_oracle as O
    from tacotron2_amd import helpers as H
    B, M, r, nf = 2, 3, 3, 2
    tgt = np.arange(B * 6 * M, dtype=np.float32).reshape(B, 6, M)
    h, ho = H.TrainingHelper(torch.from_numpy(tgt), M, r, nf), O.TrainingHelper(tgt, M, r, nf)
    fin, first = h.initialize()
    assert first.shape == (B, M * nf) and not first.any() and not fin.any()
    for t in range(2):
        f1, n1, _ = h.next_inputs(t, (None, None), None)
        f2, n2, _ = ho.next_inputs(t, (None, None), None)
        assert f1.tolist() == f2.tolist()
        np.testing.assert_array_equal(n1.numpy(), n2)
    with pytest.raises(AssertionError):
        H.TrainingHelper(torch.from_numpy(tgt), M, 2, 3)
    with pytest.raises(AssertionError):
        H.StopTokenBasedInferenceHelper(B, M, 1, 2, device="cpu")
    s = H.StopTokenBasedInferenceHelper(B, M, r, nf, min_iters=2, device="cpu")
    out = torch.ones(B, M * r)
    pos, mixed = torch.tensor([[3.0], [1.0]]), torch.tensor([[3.0], [-1.0]])
    assert not s.next_inputs(2, (out, pos), None)[0].any()        # time > min_iters is strict
    assert s.next_inputs(3, (out, pos), None)[0].all()
    assert not s.next_inputs(3, (out, mixed), None)[0].any()      # batch-coupled
    assert s.next_inputs(3, (out, pos), None)[0].shape == (1,)
    assert not s.next_inputs(3, (out, torch.zeros(B, 1)), None)[0].any()   # sigmoid(0) = 0.5 is not > 0.5
    v = H.ValidationHelper(torch.from_numpy(tgt), B, M, r, nf, teacher_forcing=False)
    f, n, _ = v.next_inputs(1, (out, pos), None)
    assert f.all() and n.shape == (B, M * nf)
    i = H.InferenceHelper(B, M, r, nf, device="cpu")
    z = torch.zeros(B, M * r)
    z[1, 0] = 1
    assert i.next_inputs(0, (z, None), None)[0].tolist() == [True, False]


def test_constructor_checks_mirror_the_reference():
    from tacotron2_amd.modules import ZoneoutLSTMCell
    from tacotron2_amd.rnn_impl import lstm_cell_factory, gru_cell_factory, LSTMImpl
    from tacotron2_amd.rnn_wrappers import OutputProjectionWrapper
    with pytest.raises(ValueError):
        ZoneoutLSTMCell(4, False, zoneout_factor_cell=1.5)
    with pytest.raises(ValueError):
        lstm_cell_factory("nope", 4)
    with pytest.raises(ValueError):
        gru_cell_factory("nope", 4)
    assert lstm_cell_factory(LSTMImpl.LSTMBlockCell, 4).state_size == (4, 4)
    with pytest.raises(ValueError):
        OutputProjectionWrapper(None, 0)
    with pytest.raises(NotImplementedError):
        ZoneoutLSTMCell(4, False, state_is_tuple=False)(None, None)


def test_sharding_covers_every_utterance_once():
    from tacotron2_amd.sharding import make_batches, shard_batches
    rng = np.random.default_rng(0)
    lens = rng.integers(60, 141, 512)
    batches = make_batches(lens, 64)
    assert len(batches) == 8 and sorted(np.concatenate(batches).tolist()) == list(range(512))
    got = []
    for rank in range(8):
        got += shard_batches(len(batches), 8, rank)
    assert sorted(got) == list(range(8))
    # length-sorted batches have little padding
    assert all(lens[b].max() - lens[b].min() <= 15 for b in batches)


def test_flat_parameter_buffer_roundtrip():
    from tacotron2_amd import training as TR
    from tacotron2_amd.hparams import PROFILE_TINY
    from tacotron2_amd.weights import init_weights
    w = init_weights(PROFILE_TINY, seed=3)
    spec = TR.flatten_spec(w)
    flat = TR.flatten(w, spec)
    assert flat.size == TR.flat_size(spec) and flat.size % 4 == 0
    assert all(off % 4 == 0 for _, _, off, _ in spec)                 # every variable float4-aligned
    assert [n for n, *_ in spec] == sorted(w)
    back = TR.unflatten(flat, spec)
    for k in w:
        np.testing.assert_array_equal(back[k], np.asarray(w[k], np.float32))
    assert abs(TR.learning_rate_decay(0.002, 3999) - 0.002) < 1e-9
