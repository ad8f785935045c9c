"""Teacher-forced persistent decode: two-chain schedule (option tf2 = 1, persist.h MODE 3) against the four-hop schedule (tf2 = 0):
same frames, time per step, no hop timeouts.  python tools/tf2_check.py [B=1] [steps=400]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bench import make_inputs
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
B, S = 1, 400
for a in sys.argv[1:]:
    if a.startswith("B="): B = int(a[2:])
    if a.startswith("steps="): S = int(a[6:])
d = PROFILE_P.replace(max_iters=S)
w = init_weights(d, seed=1234)
eng = Engine(d, w, device=0)
ids, lens = make_inputs(B, 60, 140, seed=1235)
T = int(lens.max())
ids = np.ascontiguousarray(ids[:, :T])
mem = eng.encode(ids, lens)
tgt = torch.tensor(np.random.default_rng(3).uniform(-1, 1, (B, S * d.outputs_per_step, d.num_mels)).astype(np.float32), device="cuda")
res = {}
for tf2 in (0, 1):
    eng.set_option("tf2", tf2)
    out = eng.decode_teacher(mem, lens, tgt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): out = eng.decode_teacher(mem, lens, tgt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    res[tf2] = [o.cpu().numpy() for o in out]
    print("B=%d tf2=%d: %.2f ms per %d steps = %.2f us/step; persist launches %d, timeouts %d" %
          (B, tf2, dt * 1e3, S, dt * 1e6 / S, eng.get_stat("persist_launches"), eng.get_stat("persist_timeouts")), flush=True)
for name, a, b in zip(("mel", "stop", "align"), res[0], res[1]):
    print("max |%s(tf2=1) - %s(tf2=0)| = %.3g (scale %.3g)" % (name, name, float(np.abs(a - b).max()), float(np.abs(a).max())))
eng.close()
