"""Diagnostic (not collected): per-variable gradient error table of the PostNet training step."""
import sys
import numpy as np
import torch
sys.path[:0] = ["tests", "."]
import test_gpu_v1_train as t
from oracle import v1_torch_train as VT
from tacotron2_amd import training as TR
from tacotron2_amd.tacotron_v1 import PostNetTrainer

for d, B, T in [(t.POST, 5, 23), (t.POST, 33, 40)]:
    w, eng, pw, spec, flat = t._setup(d)
    mel, target, mask = t._case(d, B, T, 100 * B + T)
    ref = VT.postnet_train_forward_backward(d, w, mel, target, mask)
    tr = PostNetTrainer(eng, B, T, spec)
    out = tr.forward(flat, mel)
    print("forward err", float(np.abs(t._np(out) - ref["spec"]).max()))
    grads = torch.full_like(flat, float("nan"))
    parts = t._np(tr.backward(target, mask, grads))
    got = TR.unflatten(grads, spec)
    rows = []
    for name, g in ref["grads"].items():
        if name.endswith(("moving_mean", "moving_variance")):
            continue
        sc = float(np.abs(g).max())
        rows.append((float(np.abs(got[name] - g).max()) / sc, sc, name))
    for hn in ("highway_1", "highway_2", "highway_3", "highway_4"):
        k = "v1/postnet/cbhg/%s/H/kernel" % hn
        e = np.abs(got[k] - ref["grads"][k]).max(axis=0)
        top = np.argsort(e)[::-1][:4]
        print(hn, "H/kernel error by column: top", [(int(c), float(e[c])) for c in top], "median", float(np.median(e)))
    print(B, T, "gmax", max(r[1] for r in rows))
    for r in sorted(rows, key=lambda r: r[2]):
        print("  %.2e  scale %.2e  %s" % r)
    tr.close(); eng.close()
