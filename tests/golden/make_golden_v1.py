#!/usr/bin/env python3
"""Generate the Tacotron v1 golden fixtures (tests/golden/v1_*.npz) with the CPU oracle (oracle/v1_oracle.py).
Same rules as make_golden.py: the reference ships no vectors, these pin the oracle's own outputs; inputs and
weights come from seeds (``cases()``), each fixture stores outputs + a weights checksum.

    python tests/golden/make_golden_v1.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import v1_oracle as V                          # noqa: E402
from tacotron2_amd.hparams import V1_PROFILE, V1_TINY      # noqa: E402
from tacotron2_amd.weights_v1 import init_weights_v1       # noqa: E402

V1_MID = V1_PROFILE.replace(num_symbols=60, embedding_dim=48, encoder_prenet_out_units=(48, 40), cbhg_out_units=72, conv_channels=20,
                            max_filter_width=5, projection1_out_channels=36, projection2_out_channels=40, num_highway=2,
                            decoder_prenet_out_units=(44, 36), attention_out_units=136, decoder_out_units=100, num_mels=20,
                            num_freq=33, post_net_cbhg_out_units=24, post_net_conv_channels=12, post_net_max_filter_width=4,
                            post_net_projection1_out_channels=28, post_net_projection2_out_channels=20, post_net_num_highway=2,
                            max_iters=40, min_iters=5)


def weights_checksum(w):
    return float(sum(float(np.sum(np.abs(v), dtype=np.float64)) for _, v in sorted(w.items())))


def tune_stop(d, w, memory, lens, lo, hi, min_gap=4e-3):
    """Stop threshold in the widest gap of the per-step batch-minimum stop logits (SURVEY.md 8c fixture 4)."""
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


def cases():
    out = {}
    rng = np.random.default_rng(301)
    d = V1_TINY
    out["v1_tiny_teacher"] = (d, init_weights_v1(d, 31, perturb=True), {
        "ids": rng.integers(0, d.num_symbols, (3, 7)), "lens": np.array([7, 4, 6]),
        "targets": rng.uniform(0, 1, (3, 10, d.num_mels)).astype(np.float32), "mode": "teacher"})
    rng = np.random.default_rng(302)
    out["v1_mid_teacher"] = (V1_MID, init_weights_v1(V1_MID, 32, perturb=True), {
        "ids": rng.integers(0, V1_MID.num_symbols, (5, 21)), "lens": np.array([21, 17, 21, 9, 14]),
        "targets": rng.uniform(0, 1, (5, 28, V1_MID.num_mels)).astype(np.float32), "mode": "teacher"})
    rng = np.random.default_rng(303)
    out["v1_mid_free"] = (V1_MID, init_weights_v1(V1_MID, 33, perturb=True), {
        "ids": rng.integers(0, V1_MID.num_symbols, (4, 16)), "lens": np.array([16, 12, 16, 10]), "mode": "free"})
    # the reference's hparams.py sizes, B=2
    rng = np.random.default_rng(304)
    out["v1_ref_teacher"] = (V1_PROFILE, None, {
        "ids": rng.integers(32, 91, (2, 24)), "lens": np.array([24, 19]),
        "targets": rng.uniform(0, 1, (2, 40, 80)).astype(np.float32), "mode": "teacher", "weight_seed": 34})
    return out


def materialize(name, case):
    d, w, inp = case
    if w is None:
        w = init_weights_v1(d, inp["weight_seed"], perturb=True)
    if inp["mode"] == "free":
        from oracle import t2_oracle as O
        memory = V.encoder_v1(d, w, O.embedding_lookup(inp["ids"], w["embedding"]), None)
        w, _ = tune_stop(d, w, memory, inp["lens"], 8, 30)
    return d, w, inp


def run_oracle(d, w, inp):
    if inp["mode"] == "teacher":
        ref = V.tacotron1_forward(d, w, inp["ids"], inp["lens"], mode="eval", target=inp["targets"], teacher_forcing=True)
    else:
        ref = V.tacotron1_forward(d, w, inp["ids"], inp["lens"], mode="predict")
    ref["spec"] = V.postnet_v1(d, w, ref["mel"])
    return ref


def main():
    for name, case in cases().items():
        d, w, inp = materialize(name, case)
        ref = run_oracle(d, w, inp)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, memory=ref["memory"], mel=ref["mel"], stop=ref["stop"], alignment=ref["alignment"],
                            spec=ref["spec"], steps=np.int64(ref["steps"]), weights_checksum=np.float64(weights_checksum(w)))
        print(name, "steps", ref["steps"], "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
