"""EncoderV1 inference forward only (profiling helper): python tools/v1_enc_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd.hparams import V1_PROFILE
from tacotron2_amd.tacotron_v1 import EngineV1
from tacotron2_amd.weights_v1 import init_weights_v1
d = V1_PROFILE
eng = EngineV1(d, init_weights_v1(d, 1239))
rng = np.random.default_rng(1)
ids = rng.integers(32, 91, (32, 140))
eng.encode(ids, None); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): eng.encode(ids, None)
torch.cuda.synchronize()
print("v1 encoder: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
eng.close()
