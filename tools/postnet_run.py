"""PostNetV2 only at the bench shape (profiling helper): python tools/postnet_run.py [name=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
d = PROFILE_P
eng = Engine(d, init_weights(d, seed=1234), device=0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
mel = torch.rand(64, 800, d.num_mels, device="cuda")
eng.postnet(mel); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): eng.postnet(mel)
torch.cuda.synchronize()
print("PostNetV2 64x800: %.3f ms %s" % ((time.perf_counter() - t0) * 100, sys.argv[1:]))
eng.close()
