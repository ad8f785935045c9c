"""Phase timing of the persistent decoder from a -DPZ_STAMPS build (tools/build_stamps.sh):
   T2MI_LIB=tools/bin/libt2mi_stamps.so python tools/persist_stamps.py B=64"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bench import make_inputs
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
B, S = 1, 400
for a in sys.argv[1:]:
    if a.startswith("B="): B = int(a[2:])
d = PROFILE_P.replace(max_iters=S)
w = init_weights(d, seed=1234)
w["decoder/stop_token_projection/bias"] = np.array([-1e3], np.float32)
eng = Engine(d, w, device=0)
ids, lens = make_inputs(B, 100, 100, seed=1235) if B == 1 else make_inputs(B, 60, 140, seed=1237)
T = ids.shape[1]
bufs = eng.decode_free_buffers(B, T, S)
mem = eng.encode(ids, lens)
for _ in range(2):
    eng.decode_free_launch(mem, lens, S, d.min_iters, bufs); n = eng.decode_free_finish()
out = np.zeros((256, 24), np.uint64)
eng.lib.t2_debug_stamps.argtypes = [C.c_void_p]
assert eng.lib.t2_debug_stamps(out.ctypes.data_as(C.c_void_p)) == 0
us = out.astype(np.float64) * 0.01 / n          # us per step per interval
names = ["P5 tail+P6 work", "arr6+gap Z2+attn pre", "WAIT6", "stop+P1 work", "arr1", "WAIT1", "attn hist+ctx", "arr2+gap Z1x", "WAIT2", "P3 work",
         "arr3+gap ZA", "WAIT3", "P4 work", "arr4+gap Z1", "WAIT4", "P5 work", "-", "-", "attn q+energies", "-", "attn exchange+softmax"]
MB = 1 if B <= 16 else 2 if B <= 32 else 4
wn = 255 - 27 * MB - 1          # an owner of a last-PreNet tile
print("B=%d T=%d steps=%d ; us per step: mean over workgroups / min / max / wg0 / wg%d (PreNet tile) / wg255 (projection tile)" % (B, T, n, wn))
for i, nm in enumerate(names):
    c = us[:, i]
    print("%2d %-22s %7.2f %7.2f %7.2f %7.2f %7.2f %7.2f" % (i, nm, c.mean(), c.min(), c.max(), c[0], c[wn], c[255]))
print("sum (mean) %.2f" % us[:, :21].sum(axis=1).mean())
eng.close()
