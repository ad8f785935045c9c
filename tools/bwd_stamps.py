"""Phase timing of the persistent backward recurrence from a -DPZ_STAMPS build (tools/build_stamps.sh):
   T2MI_LIB=tools/bin/libt2mi_stamps.so python tools/bwd_stamps.py B=32"""
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
    if a.startswith("S="): S = int(a[2:])
    if a.startswith("T="): T = int(a[2:])
d = PROFILE_P
w = init_weights(d, seed=5, perturb=True)
rng = np.random.default_rng(3)
r = d.outputs_per_step
lens = rng.integers(60, T + 1, B).astype(np.int64); lens[0] = T
memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
mel = rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32)
done = np.zeros((B, S), np.float32); done[:, -1] = 1
eng = Engine(d, w)
tr = TR.Trainer(eng, B, T, S)
spec = TR.flatten_spec(w)
tr.bind_gradients(spec)
flat = torch.zeros(TR.flat_size(spec), dtype=torch.float32, device="cuda")
tr.decoder_forward(memory, lens, mel, prenet_drop_rate=0.5, seed=3)
for _ in range(2):
    tr.decoder_backward(done, np.ones((B, S * r), np.float32), np.ones((B, S), np.float32), flat)
torch.cuda.synchronize()
out = np.zeros((256, 24), np.uint64)
if not hasattr(eng.lib, "t2_debug_stamps"):       # product build (profiling runs): the recurrences ran, there are no stamps to print
    print("product build: no phase stamps")
    tr.close(); eng.close(); sys.exit(0)
eng.lib.t2_debug_stamps.argtypes = [C.c_void_p]
assert eng.lib.t2_debug_stamps(out.ctypes.data_as(C.c_void_p)) == 0
us = out.astype(np.float64) * 0.01 / S
names = ["arrive A + gap A (GEMM0)", "cells L2,L1", "WAIT A", "GEMM1+dctx", "WAIT B", "attention main", "WAIT C", "query+cell att", "WAIT D", "arrive B + gap B (GEMM2a, att pre)", "arrive C + gap C (GEMM2b)", "gap D rest (tape prefetch)", "att main: dctx+dot", "att main: da+de", "att tail: df (+arrive D)", "att tail: ds"]
print("B=%d T=%d S=%d ; us per step: mean over workgroups / min / max / wg0 / wg255" % (B, T, S))
for i, nm in enumerate(names):
    c = us[:, i]
    print("%2d %-36s %7.2f %7.2f %7.2f %7.2f %7.2f" % (i, nm, c.mean(), c.min(), c.max(), c[0], c[255]))
print("sum (mean) %.2f" % us[:, :16].sum(axis=1).mean())
tr.close(); eng.close()
