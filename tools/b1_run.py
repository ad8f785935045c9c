"""Batch-1 free-running decode only (profiling helper): python tools/b1_run.py [name=value ...]"""
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
d = PROFILE_P.replace(max_iters=400)
w = init_weights(d, seed=1234)
w["decoder/stop_token_projection/bias"] = np.array([-1e3], np.float32)
eng = Engine(d, w, device=0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
ids, lens = make_inputs(B, 100, 100, seed=1235)
bufs = eng.decode_free_buffers(B, 100, 400)
mem = eng.encode(ids, lens)
def dec():
    eng.decode_free_launch(mem, lens, 400, d.min_iters, bufs)
    return eng.decode_free_finish()
dec(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): n = dec()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print("B=%d decode only: %.2f ms per %d steps = %.2f us/step" % (B, dt * 1e3, n, dt * 1e6 / n))
eng.close()
