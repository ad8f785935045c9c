"""Option combinations of the training recurrences: eager forward + persistent backward, and the fallback to the
launch-per-phase backward when the persistent kernel's LDS does not fit (B = 64, T_in = 256)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tacotron2_amd import training as TR
from tacotron2_amd.engine import Engine
from tacotron2_amd.hparams import PROFILE_P
from tacotron2_amd.weights import init_weights
d = PROFILE_P
w = init_weights(d, seed=5, perturb=True)
def run(B, T, S, persist_fwd, bwd):
    rng = np.random.default_rng(7)
    r = d.outputs_per_step
    lens = rng.integers(max(1, T // 2), T + 1, B).astype(np.int64); lens[0] = T
    memory = rng.uniform(-1, 1, (B, T, d.enc_units)).astype(np.float32)
    mel = rng.uniform(0, 1, (B, S * r, d.num_mels)).astype(np.float32)
    done = np.zeros((B, S), np.float32); done[:, -1] = 1
    eng = Engine(d, w); tr = TR.Trainer(eng, B, T, S)
    spec = TR.flatten_spec(w); tr.bind_gradients(spec)
    eng.set_option("persist", persist_fwd); eng.set_option("bwd_persist", bwd)
    n0 = eng.get_stat("bwd_persist_launches"); p0 = eng.get_stat("persist_launches")
    tr.decoder_forward(memory, lens, mel, prenet_drop_rate=0.5, seed=3)
    flat = torch.zeros(TR.flat_size(spec), dtype=torch.float32, device="cuda")
    dmem, losses = tr.decoder_backward(done, np.ones((B, S * r), np.float32), np.ones((B, S), np.float32), flat)
    torch.cuda.synchronize()
    out = (TR.unflatten(flat, spec), dmem.cpu().numpy(), eng.get_stat("bwd_persist_launches") - n0, eng.get_stat("persist_launches") - p0)
    tr.close(); eng.close()
    return out
def cmp(a, b):
    worst = 0.0
    for k in a[0]:
        if k.startswith("decoder") or k.startswith("attention"):
            sc = float(np.abs(a[0][k]).max()); worst = max(worst, float(np.abs(a[0][k] - b[0][k]).max()) / (sc + 1e-9))
    worst = max(worst, float(np.abs(a[1] - b[1]).max()) / (float(np.abs(a[1]).max()) + 1e-9))
    return worst
ref = run(20, 77, 5, 1, 0)
x = run(20, 77, 5, 0, 1)
print("eager forward + persistent backward: launches fwd %d bwd %d, worst rel diff %.2e" % (x[3], x[2], cmp(ref, x)))
ref = run(64, 256, 2, 1, 0)
x = run(64, 256, 2, 1, 1)
print("B=64 T=256 (LDS beyond the limit -> fallback): persistent bwd launches %d, worst rel diff %.2e" % (x[2], cmp(ref, x)))
