"""Profiling driver: N PostNet training steps (TacotronV1PostNetModel.model_fn(TRAIN)) at B=32, T=800, reference hparams sizes.
Run under `rocprofv3 --kernel-trace --stats -- python3 tools/postnet_train_prof.py`."""
import sys
import time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tacotron2_amd.hparams import V1_PROFILE
from tacotron2_amd.tacotron_v1 import EngineV1, ModeKeys, TacotronV1PostNetModel
from tacotron2_amd.weights_v1 import init_weights_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
d = V1_PROFILE
w = init_weights_v1(d, 1239)
eng = EngineV1(d, w)
model = TacotronV1PostNetModel(d, eng, weights=w)
rng = np.random.default_rng(1)
B, T = 32, 800
mel = torch.tensor(rng.uniform(0, 1, (B, T, d.num_mels)).astype(np.float32), device="cuda")
spec = torch.tensor(rng.uniform(0, 1, (B, T, d.num_freq)).astype(np.float32), device="cuda")
lens = rng.integers(T // 2, T + 1, B); lens[0] = T
mask = torch.tensor((np.arange(T)[None, :] < lens[:, None]).astype(np.float32), device="cuda")
model.model_fn({"mel": mel}, {"spec": spec, "spec_loss_mask": mask}, ModeKeys.TRAIN)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    res = model.model_fn({"mel": mel}, {"spec": spec, "spec_loss_mask": mask}, ModeKeys.TRAIN)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / n * 1e3, "loss", float(res["loss"]))
eng.close()
