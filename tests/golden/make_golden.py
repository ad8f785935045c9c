#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the CPU oracle (oracle/t2_oracle.py).

The reference ships no golden vectors and TensorFlow cannot run in this pipeline, so these vectors
pin the *oracle's* outputs (parity with TensorFlow itself stays unpinned, see oracle/t2_oracle.py).
Inputs and weights are regenerated from seeds by ``cases()``; each fixture stores the expected
outputs plus a checksum of the weights so that a drift of the seeded generators is detected.

    python tests/golden/make_golden.py            # rewrite every fixture
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import t2_oracle as O                      # noqa: E402
from tacotron2_amd.hparams import PROFILE_P, PROFILE_TINY, T2Dims      # noqa: E402
from tacotron2_amd.weights import init_weights, sharpen_attention         # noqa: E402

MID = T2Dims(num_symbols=64, num_mels=20, outputs_per_step=2, n_feed_frame=2, attention_kernel=31,
             attention_filters=32, embedding_dim=64, enc_conv_layers=3, enc_kernel=5, enc_units=64,
             prenet_units=(48, 32), attention_lstm_units=96, decoder_lstm_units=80, attention_dim=32,
             postnet_layers=5, postnet_kernel=5, postnet_channels=64, max_iters=40, min_iters=10)


def weights_checksum(w):
    return float(sum(float(np.sum(np.abs(v), dtype=np.float64)) for _, v in sorted(w.items())))


def cases():
    """name -> (dims, weights, inputs dict).  Deterministic."""
    out = {}
    for r, nf in ((1, 1), (2, 1), (2, 2)):
        d = PROFILE_TINY.replace(outputs_per_step=r, n_feed_frame=nf)
        rng = np.random.default_rng(100 + 10 * r + nf)
        out[f"tiny_r{r}_nf{nf}"] = (d, init_weights(d, 21, perturb=True), {
            "ids": rng.integers(0, d.num_symbols, (3, 7)), "lens": np.array([7, 5, 3]),
            "targets": rng.uniform(0, 1, (3, 6 * r, d.num_mels)).astype(np.float32), "mode": "teacher"})
    rng = np.random.default_rng(7)
    out["mid_teacher"] = (MID, init_weights(MID, 5, perturb=True), {
        "ids": rng.integers(0, MID.num_symbols, (6, 23)), "lens": np.array([23, 20, 17, 23, 12, 15]),
        "targets": rng.uniform(0, 1, (6, 30, MID.num_mels)).astype(np.float32), "mode": "teacher"})
    # free-running decode with a stop-tuned projection (SURVEY.md section 8c fixture 4)
    rng = np.random.default_rng(12)
    w = init_weights(MID, 9, perturb=True)
    ids = rng.integers(0, MID.num_symbols, (4, 19))
    lens = np.array([19, 14, 19, 11])
    out["mid_free"] = (MID, w, {"ids": ids, "lens": lens, "mode": "free"})
    # paper sizes, BASELINE.json configs[0]: B=1, T_in=100 (99 ids in [32,90] + eos=1), 800 target frames
    rng = np.random.default_rng(1234 + 1)
    idp = np.concatenate([rng.integers(32, 91, 99), [1]])[None, :]
    out["paper_b1_teacher"] = (PROFILE_P, None, {          # weights: init_weights(PROFILE_P, 1235), made lazily (150 MB)
        "ids": idp, "lens": np.array([100]),
        "targets": rng.uniform(0, 1, (1, 800, 80)).astype(np.float32), "mode": "teacher", "weight_seed": 1235})
    # the same case with a PEAKED attention (weights.sharpen_attention): with Glorot-initialised attention weights the
    # alignments are uniform to ~1e-3 and no tolerance can see the terms of the score (tests/test_oracle_sensitivity.py)
    out["paper_sharp_b1_teacher"] = (PROFILE_P, None, dict(out["paper_b1_teacher"][2], sharp=True))
    # BASELINE.json configs[2] shape: B = 64 free-running rows, ragged T_in in 60..140, peaked attention, the stop
    # projection tuned so that the batch-coupled rule (helpers.py:90-94) fires at a step in [24, 40)
    rng = np.random.default_rng(1234 + 2)
    lens = rng.integers(60, 141, size=64).astype(np.int64)
    lens[7] = 140
    ids = np.zeros((64, 140), np.int64)
    for b, n in enumerate(lens):
        ids[b, :n - 1] = rng.integers(32, 91, size=n - 1)
        ids[b, n - 1] = 1
    out["paper_sharp_b64_free"] = (PROFILE_P.replace(max_iters=40), None, {
        "ids": ids, "lens": lens, "mode": "free", "weight_seed": 1236, "sharp": True, "perturb": True,
        "stop_gain": 25.0, "stop_window": (24, 40)})
    return out


def tune_stop(d, w, memory, lens, lo, hi, min_gap=4e-3):
    """Put the stop threshold into the widest gap of the per-step batch-minimum stop logits so that
    the batch-coupled rule fires at a step in [lo, hi) and every step's decision has a margin."""
    best = None
    for sign in (1.0, -1.0):
        w2 = dict(w)
        w2["decoder/stop_token_projection/kernel"] = (sign * w["decoder/stop_token_projection/kernel"]).astype(np.float32)
        w2["decoder/stop_token_projection/bias"] = np.zeros(1, np.float32)
        _, stop, _ = O.decoder_v2(d.replace(min_iters=hi + 1), w2, memory, memory_sequence_length=lens, max_iters=hi)
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
    w2 = best[3]
    w2["decoder/stop_token_projection/bias"] = np.array([-best[2]], np.float32)
    return w2, best[0] + 1


def materialize(name, case):
    d, w, inp = case
    if w is None:
        w = init_weights(d, inp["weight_seed"], perturb=inp.get("perturb", False))
    if inp.get("sharp"):
        w = sharpen_attention(w)
    if "stop_gain" in inp:      # a random-init stop logit drifts by ~5e-3 per step at profile P: too little for a margin
        w = dict(w)
        w["decoder/stop_token_projection/kernel"] = (w["decoder/stop_token_projection/kernel"] * inp["stop_gain"]).astype(np.float32)
    if inp["mode"] == "free":
        memory = O.encoder_v2(d, w, O.embedding_lookup(inp["ids"], w["embedding"]), inp["lens"])
        lo, hi = inp.get("stop_window", (15, 36))
        w, _ = tune_stop(d, w, memory, inp["lens"], lo, hi, min_gap=inp.get("min_gap", 4e-3))
    return d, w, inp


def run_oracle(d, w, inp):
    if inp["mode"] == "teacher":
        return O.tacotron2_forward(d, w, inp["ids"], inp["lens"], mode="eval", target=inp["targets"], teacher_forcing=True)
    return O.tacotron2_forward(d, w, inp["ids"], inp["lens"], mode="predict")


def fp64_noise(d, w, inp, ref):
    """Conditioning record of a fixture: the oracle re-run in float64 (same code, O.F32 swapped) and the per-decoder-step
    max |fp32 - fp64| of alignments, mel and stop.  A peaked attention with cumulative location feedback has windows in
    which rounding noise is amplified ~100 x (and decays again); a test tolerance must sit above this floor there."""
    O.F32 = np.float64
    try:
        r64 = run_oracle(d, {k: v.astype(np.float64) for k, v in w.items()}, inp)
    finally:
        O.F32 = np.float32
    S = ref["steps"]
    assert r64["steps"] == S
    B, r = ref["mel"].shape[0], d.outputs_per_step
    a = np.abs(ref["alignment"] - r64["alignment"]).max((0, 1))
    m = np.abs(ref["mel"] - r64["mel"]).reshape(B, S, -1).max((0, 2))
    st = np.abs(ref["stop"] - r64["stop"]).reshape(B, S).max(0)
    return a.astype(np.float32), m.astype(np.float32), st.astype(np.float32)


def main():
    only = set(sys.argv[1:])
    for name, case in cases().items():
        if only and name not in only:
            continue
        d, w, inp = materialize(name, case)
        ref = run_oracle(d, w, inp)
        path = os.path.join(HERE, name + ".npz")
        extra = {}
        if inp.get("sharp"):
            a_n, m_n, s_n = fp64_noise(d, w, inp, ref)
            extra = {"alignment_noise": a_n, "mel_noise": m_n, "stop_noise": s_n}
            print("   fp32-vs-fp64 noise of the oracle: alignment max %.1e median %.1e, mel max %.1e, stop max %.1e"
                  % (a_n.max(), np.median(a_n), m_n.max(), s_n.max()))
        rows = np.arange(ref["memory"].shape[0])
        if ref["memory"].nbytes > (4 << 20):        # B = 64 at profile P: 18 MB of encoder output; keep a sample of the rows
            rows = np.array([0, 7, 31, 63])
        np.savez_compressed(path, memory=ref["memory"][rows].astype(np.float32), memory_rows=rows, mel=ref["mel"], stop=ref["stop"],
                            alignment=ref["alignment"], mel_postnet=ref["mel_postnet"], steps=np.int64(ref["steps"]),
                            weights_checksum=np.float64(weights_checksum(w)),
                            stop_bias=w["decoder/stop_token_projection/bias"],
                            stop_kernel=w["decoder/stop_token_projection/kernel"], **extra)
        peak = np.transpose(ref["alignment"], (2, 0, 1)).max(-1)          # [S, B]
        print(name, "steps", ref["steps"], "alignment peak mean %.3f min %.3f" % (peak.mean(), peak.min()),
              "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
