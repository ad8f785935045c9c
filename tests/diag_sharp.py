"""Diagnostic (not collected by pytest): per-step error of the HIP decoder against the sharp-attention golden fixtures, next to
the oracle's own fp32-vs-fp64 noise.  Run on the GPU box: python tests/diag_sharp.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden import make_golden as G
from tacotron2_amd.engine import Engine

np.set_printoptions(precision=1, linewidth=220, threshold=10000)
for name in ("paper_sharp_b1_teacher", "paper_sharp_b64_free"):
    d, w, inp = G.materialize(name, G.cases()[name])
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    eng = Engine(d, w)
    for persist in (1, 0):
        eng.set_option("persist", persist)
        memory = eng.encode(inp["ids"], inp["lens"])
        if inp["mode"] == "teacher":
            mel, stop, align = eng.decode_teacher(memory, inp["lens"], inp["targets"], teacher_forcing=True)
        else:
            mel, stop, align, n = eng.decode_free(memory, inp["lens"])
            print("steps", n, int(gold["steps"]))
        a = align.permute(1, 2, 0).cpu().numpy()
        S = min(a.shape[2], gold["alignment"].shape[2])
        da = np.abs(a[:, :, :S] - gold["alignment"][:, :, :S]).max((0, 1))
        noise = gold["alignment_noise"][:S]
        print(name, "persist", persist, "align err max %.2e; max err/noise ratio %.1f; mel err %.2e stop err %.2e" % (
            da.max(), (da / np.maximum(noise, 1e-6)).max(), np.abs(mel.cpu().numpy()[:, :S * 2] - gold["mel"][:, :S * 2]).max(),
            np.abs(stop.cpu().numpy()[:, :S] - gold["stop"][:, :S]).max()))
        print(" err  ", da[::4] if S > 100 else da)
        print(" noise", noise[::4] if S > 100 else noise)
    eng.close()
