"""Profile-P (paper-size) inputs shared by the GPU parity tests and by the CPU meta-test that proves those tests can fail
(tests/test_oracle_sensitivity.py): the same seeds give the same weights, ids and lengths on both sides."""
import numpy as np

from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights, sharpen_attention

ATOL_PAPER = 2e-4            # what the paper-size GPU tests allow on alignments / mel / stop (north_star: 1e-3)


def paper_case(weight_seed, input_seed, sharp, B=64, full_row=2):
    """(d, w, ids [B, T], lens [B]): ragged T_in in 60..140 with one full-length row, ids in the LJSpeech range."""
    d = PROFILE_P
    w = init_weights(d, seed=weight_seed, perturb=True)
    if sharp:
        w = sharpen_attention(w)
    rng = np.random.default_rng(input_seed)
    lens = rng.integers(60, 141, size=B).astype(np.int64)
    lens[full_row] = 140
    T = int(lens.max())
    ids = np.zeros((B, T), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(32, 91, size=n)
    return d, w, ids, lens


ATTENTION_TERMS = ("attention/location_features_layer/kernel", "attention/location_features_convolution/kernel",
                   "attention/query_layer/kernel", "attention/attention_bias")


def train_case(B, T, S, weight_seed=5, input_seed=1, sharp=True):
    """A training batch at profile P the way datasets.py pads it: ragged source / target lengths, done flags, loss masks."""
    d = PROFILE_P
    w = init_weights(d, seed=weight_seed, perturb=True)
    if sharp:
        w = sharpen_attention(w)
    rng = np.random.default_rng(input_seed)
    r = d.outputs_per_step
    ids = rng.integers(0, d.num_symbols, (B, T))
    lens = rng.integers(max(1, (3 * T) // 7), T + 1, B).astype(np.int64)
    lens[0] = T
    tl = rng.integers(S * r // 2, S * r + 1, B)
    tl[0] = S * r
    sl = (tl + r - 1) // r
    labels = {"mel": rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32),
              "done": (np.arange(S)[None, :] >= (sl[:, None] - 1)).astype(np.float32),
              "spec_loss_mask": (np.arange(S * r)[None, :] < tl[:, None]).astype(np.float32),
              "binary_loss_mask": (np.arange(S)[None, :] < sl[:, None]).astype(np.float32)}
    return d, w, ids, lens, labels
