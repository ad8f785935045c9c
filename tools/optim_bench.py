"""HBM roofline of the optimizer block (T4) and the losses (T1/T2) at profile-P sizes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd.engine import shared_context
from tacotron2_amd import training as TR
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import weight_shapes

ctx = shared_context(0)
n = sum(((int(np.prod(s)) + 3) & ~3) for s in weight_shapes(PROFILE_P).values())
p = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda") * 1e-3
opt = TR.AdamOptimizer(ctx, n, learning_rate=1e-3)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
us = timed(lambda: opt.apply_gradients(g, p))
byts = n * 4 * (1 + 7)          # norm pass reads g; Adam reads g,m,v,p and writes m,v,p
print("adam+clip: n=%d params, %.1f us/step, %.0f GB/s algorithmic (%.2f of 8 TB/s)" % (n, us, byts / us / 1e3, byts / us / 1e3 / 8000))
B, T, M = 32, 800, 80
yh, y, mk = torch.randn(B, T, M, device="cuda"), torch.rand(B, T, M, device="cuda"), torch.ones(B, T, device="cuda")
us = timed(lambda: TR.spec_loss_parts(ctx, yh, y, mk))
print("spec_loss [%d,%d,%d]: %.1f us, %.0f GB/s" % (B, T, M, us, 2 * B * T * M * 4 / us / 1e3))
