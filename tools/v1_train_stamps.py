"""Where a stage slot of the V1 trainer's persistent recurrences goes (body / arrive / wait), from the diagnostic build:
bash tools/build_v1stamps.sh && T2MI_LIB=tools/bin/libt2mi_v1stamps.so python tools/v1_train_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd import _lib
from tacotron2_amd.hparams import V1_PROFILE
from tacotron2_amd.tacotron_v1 import ModeKeys, SingleSpeakerTacotronV1Model
from tacotron2_amd.weights_v1 import init_weights_v1

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
lib = _lib.load()
lib.t2v1_debug_stamps.restype = C.c_int
lib.t2v1_debug_stamps.argtypes = [C.c_void_p, C.c_int]
model.model_fn(feats, labels, ModeKeys.TRAIN)
torch.cuda.synchronize()
lib.t2v1_debug_stamps(None, 1)
n = 3
for _ in range(n):
    model.model_fn(feats, labels, ModeKeys.TRAIN)
torch.cuda.synchronize()
out = (C.c_ulonglong * 72)()
lib.t2v1_debug_stamps(out, 0)
body = np.array(list(out)[64:68], dtype=np.float64) * 0.01 / (n * S)
a = np.array(list(out)[:64], dtype=np.float64).reshape(2, 2, 16) * 0.01 / (n * S)       # us per decoder step
names = [["decoder chain (oproj, g1, c1, g2, c2)", "attention chain (ga, ca, q, scores, softmax + context)"],
         ["decoder chain (2.2, 2.3, 1.2, 1.3, -)", "attention chain (oproj', attn', q'+1, a.2, a.3)"]]
for k, kn in enumerate(("forward", "backward")):
    for g in range(2):
        print("%s, %s: total %.1f us per step" % (kn, names[k][g], a[k, g, :15].sum()))
        for sl_ in range(5):
            print("   slot %d: body %5.2f  arrive %5.2f  wait %5.2f" % (sl_, a[k, g, 3 * sl_], a[k, g, 3 * sl_ + 1], a[k, g, 3 * sl_ + 2]))
print("product stages of workgroup 0 of both halves, us per decoder step (17 stages): loads issued %.2f, K loop %.2f, reduction barrier %.2f, epilogue %.2f" % tuple(body))
model.close()
