"""Teacher-forced decode only (profiling helper): python tools/tf_run.py [B=n] [name=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bench import make_inputs
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
B = 1
for a in list(sys.argv[1:]):
    if a.startswith("B="):
        B = int(a[2:]); sys.argv.remove(a)
d = PROFILE_P
eng = Engine(d, init_weights(d, seed=1234), device=0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
ids, lens = make_inputs(B, 100, 100, seed=1235)
S = 400
tgt = torch.rand(B, S * d.outputs_per_step, d.num_mels, device="cuda")
mem = eng.encode(ids, lens)
eng.decode_teacher(mem, lens, tgt, teacher_forcing=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.decode_teacher(mem, lens, tgt, teacher_forcing=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("B=%d teacher-forced decode: %.2f ms per %d steps = %.2f us/step" % (B, dt * 1e3, S, dt * 1e6 / S))
eng.close()
