"""EncoderV2 only (profiling helper): python tools/enc_run.py [B=n] [name=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bench import make_inputs
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
B = 64
for a in list(sys.argv[1:]):
    if a.startswith("B="):
        B = int(a[2:]); sys.argv.remove(a)
d = PROFILE_P
eng = Engine(d, init_weights(d, seed=1234), device=0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
ids, lens = make_inputs(B, 60, 140, seed=1237) if B > 1 else make_inputs(1, 100, 100, seed=1235)
eng.encode(ids, lens); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): eng.encode(ids, lens)
torch.cuda.synchronize()
print("B=%d T=%d encode: %.3f ms %s" % (B, ids.shape[1], (time.perf_counter() - t0) * 100, sys.argv[1:]))
eng.close()
