"""SingleSpeakerTacotronV1Model.model_fn(TRAIN) at the reference's sizes (hparams.py: batch_size 32, r = 2), T_in <= 140, 800 frames:
python tools/v1_train_bench.py [steps]   (profiling helper; run under rocprofv3 --kernel-trace --stats for the kernel split)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd.hparams import V1_PROFILE
from tacotron2_amd.tacotron_v1 import ModeKeys, SingleSpeakerTacotronV1Model
from tacotron2_amd.weights_v1 import init_weights_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = V1_PROFILE
model = SingleSpeakerTacotronV1Model(d, weights=init_weights_v1(d, 1239), seed=7)
rng = np.random.default_rng(1)
B, T, To = 32, 140, 800
r = d.outputs_per_step
S = To // r
ids = rng.integers(32, 91, (B, T))
lens = rng.integers(60, T + 1, B); lens[0] = T
tl = rng.integers(To // 2, To + 1, B); tl[0] = To
sl = (tl + r - 1) // r
labels = {"mel": torch.tensor(rng.uniform(0, 1, (B, To, d.num_mels)).astype(np.float32), device="cuda"),
          "done": (np.arange(S)[None, :] >= (sl[:, None] - 1)).astype(np.float32),
          "spec_loss_mask": (np.arange(To)[None, :] < tl[:, None]).astype(np.float32),
          "binary_loss_mask": (np.arange(S)[None, :] < sl[:, None]).astype(np.float32)}
feats = {"source": ids, "source_length": lens}
res = model.model_fn(feats, labels, ModeKeys.TRAIN)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    res = model.model_fn(feats, labels, ModeKeys.TRAIN)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("V1 training step: %.1f ms = %.0f mel frames/s, loss %.4f" % (dt * 1e3, B * To / dt, float(res["loss"])))
model.close()
