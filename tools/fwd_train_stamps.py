"""Phase timing of the persistent training forward (pz::decode_kernel<MB, 2>) from a -DPZ_STAMPS build:
   T2MI_LIB=tools/bin/libt2mi_stamps.so python tools/fwd_train_stamps.py B=32"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd import training as TR
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
B, S, T = 32, 400, 140
for a in sys.argv[1:]:
    if a.startswith("B="): B = int(a[2:])
d = PROFILE_P
w = init_weights(d, seed=5, perturb=True)
rng = np.random.default_rng(3)
r = d.outputs_per_step
lens = rng.integers(60, T + 1, B).astype(np.int64); lens[0] = T
memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
mel = rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32)
eng = Engine(d, w)
tr = TR.Trainer(eng, B, T, S)
for _ in range(2):
    tr.decoder_forward(memory, lens, mel, prenet_drop_rate=0.5, seed=3)
torch.cuda.synchronize()
out = np.zeros((256, 24), np.uint64)
eng.lib.t2_debug_stamps.argtypes = [C.c_void_p]
assert eng.lib.t2_debug_stamps(out.ctypes.data_as(C.c_void_p)) == 0
us = out.astype(np.float64) * 0.01 / S
names = ["-", "-", "-", "P1 work", "arr1+gap", "WAIT1", "attn hist+ctx", "arr2+gap Z1x", "WAIT2", "P3 work",
         "arr3+gap ZA", "WAIT3", "P4 work", "arr4+gap Z1+attn pre", "WAIT4", "(loop tail)", "-", "-", "attn q+energies", "-", "attn exchange+softmax"]
print("B=%d T=%d S=%d ; us per step: mean over workgroups / min / max / wg0 / wg255" % (B, T, S))
for i, nm in enumerate(names):
    c = us[:, i]
    if nm != "-": print("%2d %-24s %7.2f %7.2f %7.2f %7.2f %7.2f" % (i, nm, c.mean(), c.min(), c.max(), c[0], c[255]))
print("sum (mean) %.2f" % us[:, :21].sum(axis=1).mean())
tr.close(); eng.close()
