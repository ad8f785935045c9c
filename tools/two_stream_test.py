"""Throughput experiment: two independent B=64 batches in flight on one GPU (two engines / streams)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
from bench import make_inputs

d = PROFILE_P.replace(max_iters=400)
w = init_weights(d, 1234); w["decoder/stop_token_projection/bias"] = np.array([-1e3], np.float32)
NE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engs = [Engine(d, w) for _ in range(NE)]
B, S, r = 64, 400, 2
inp = [make_inputs(B, 60, 140, 1237 + i) for i in range(NE)]
bufs = [e.decode_free_buffers(B, inp[i][0].shape[1], S) for i, e in enumerate(engs)]

def round_():
    mems = [e.encode(*inp[i]) for i, e in enumerate(engs)]
    for i, e in enumerate(engs):
        e.decode_free_launch(mems[i], inp[i][1], S, d.min_iters, bufs[i])
    outs = []
    for i, e in enumerate(engs):
        n = e.decode_free_finish()
        outs.append(e.postnet(bufs[i][0][:, :n * r]))
    return outs

round_(); torch.cuda.synchronize()
t0 = time.perf_counter(); K = 3
for _ in range(K): round_()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"engines={NE}: {NE*K*B*S*r/dt:.0f} frames/s, {dt/K/NE*1e3:.1f} ms per batch")
