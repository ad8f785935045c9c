"""Meta-test (CPU): the paper-size parity tests must be ABLE to fail.

With Glorot-initialised attention weights at profile P the softmax over T_in positions is uniform to ~1e-3, and dropping a
whole term of the location-sensitive score (tacotron_v2.py:66-78,118-141) moves the alignments by about the tolerance the
GPU tests use -- a kernel without that term would pass.  The GPU tests therefore run on weights.sharpen_attention(...);
this file holds, on the oracle, the property that makes them meaningful: removing any one term of the score, or using the
non-cumulative attention state, moves the alignments by at least 10 x the GPU tolerance on exactly the inputs the GPU tests
use; and the attention gradients of the training step are far above the gradient tests' absolute floor."""
import numpy as np
import pytest

from oracle import t2_oracle as O
from tests.paper_cases import ATOL_PAPER, ATTENTION_TERMS, paper_case

ROWS = 4            # the first rows of the B = 64 batch the GPU tests decode (B = 1 and B = 5 are sub-batches of them)


def _alignments(d, w, mem, lens, S):
    mel, stop, st = O.decoder_v2(d.replace(min_iters=100), w, mem, memory_sequence_length=lens, max_iters=S)
    return np.stack(st[0].alignment_history, 0), mel, stop


@pytest.mark.parametrize("weight_seed,input_seed,full_row", [(91, 65, 2), (77, 64, 5)])      # test_gpu_persist / test_gpu_fullsize
def test_every_attention_term_is_visible_at_the_gpu_tolerance(weight_seed, input_seed, full_row):
    d, w, ids, lens = paper_case(weight_seed, input_seed, sharp=True, full_row=full_row)
    ids, lens = ids[:ROWS], lens[:ROWS]
    mem = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    S = 6
    base, mel, stop = _alignments(d, w, mem, lens, S)
    peak = base.max(-1)
    assert peak.mean() > 0.5, "attention is not peaked"
    moves = sum(len(set(base[:, b].argmax(-1).tolist())) - 1 for b in range(ROWS))
    assert moves >= 3, "the arg-max does not move"
    for name in ATTENTION_TERMS:
        w2 = dict(w)
        w2[name] = np.zeros_like(w[name])
        a2, mel2, _ = _alignments(d, w2, mem, lens, S)
        for b in range(ROWS):                                # every row on its own: B = 1 sub-batches are tested too
            assert np.abs(a2[:, b] - base[:, b]).max() >= 10 * ATOL_PAPER, (name, b)
        assert np.abs(mel2 - mel).max() >= 5 * ATOL_PAPER, name
    a2, _, _ = _alignments(d.replace(cumulative_weights=False), w, mem, lens, S)
    assert np.abs(a2 - base).max() >= 10 * ATOL_PAPER, "cumulative_weights"


def test_uniform_attention_would_hide_the_location_term():
    """Why the sharp weights exist: the record of the judge's finding as an executable fact."""
    d, w, ids, lens = paper_case(91, 65, sharp=False)
    ids, lens = ids[:1], lens[:1]
    mem = O.encoder_v2(d, w, O.embedding_lookup(ids, w["embedding"]), lens)
    base, _, _ = _alignments(d, w, mem, lens, 6)
    w2 = dict(w)
    w2["attention/location_features_layer/kernel"] = np.zeros_like(w["attention/location_features_layer/kernel"])
    a2, _, _ = _alignments(d, w2, mem, lens, 6)
    assert np.abs(a2 - base).max() < 10 * ATOL_PAPER


def test_attention_gradients_are_far_above_the_gradient_floor():
    """The gradient tests accept err <= 2e-3 * scale + 1e-9 per tensor: every trainable tensor's scale must be >= 1e-6 so that
    the absolute floor is <= 1e-3 of it (with Glorot attention weights four attention gradients sit at 2e-7 .. 2.5e-6)."""
    from oracle import torch_train as TT
    from tests.paper_cases import train_case
    d, w, ids, lens, labels = train_case(B=6, T=50, S=3)
    ref = TT.train_forward_backward(d, w, ids, lens, labels, seed=3, prenet_drop_rate=0.5, encoder_drop_rate=0.5)
    for name, g in ref["grads"].items():
        if "moving_" in name:
            continue
        assert float(np.abs(g).max()) >= 1e-6, name
