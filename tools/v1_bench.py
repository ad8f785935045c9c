"""BASELINE.json configs[4]: Tacotron v1, B=32, reference hparams sizes: teacher-forced mel frames/s
(encoder + decoder) and PostNet (mel -> linear) spectrogram frames/s.  Synthetic inputs, random-init weights."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd.hparams import V1_PROFILE
from tacotron2_amd.tacotron_v1 import EngineV1
from tacotron2_amd.weights_v1 import init_weights_v1

d = V1_PROFILE
eng = EngineV1(d, init_weights_v1(d, 1234 + 5))
rng = np.random.default_rng(1234 + 5)
B, T_in, T_out = 32, 140, 800
lens = rng.integers(60, T_in + 1, B); lens[0] = T_in
ids = rng.integers(32, 91, (B, T_in))
targets = torch.tensor(rng.uniform(0, 1, (B, T_out, d.num_mels)).astype(np.float32), device="cuda")

def fwd():
    mem = eng.encode(ids, None)
    return eng.decode_teacher(mem, lens, targets, teacher_forcing=True)[0]

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out

dt, mel = timed(fwd)
print("v1 teacher-forced forward B=%d T_in=%d T_out=%d: %.1f ms -> %.0f mel frames/s (%.1f us per decoder step)"
      % (B, T_in, T_out, dt * 1e3, B * T_out / dt, dt * 1e6 / (T_out // d.outputs_per_step)))
dt, spec = timed(lambda: eng.postnet(mel))
print("v1 postnet B=%d T=%d: %.1f ms -> %.0f spectrogram frames/s" % (B, T_out, dt * 1e3, B * T_out / dt))
dt, _ = timed(lambda: eng.encode(ids, None))
print("v1 encoder B=%d T_in=%d: %.2f ms" % (B, T_in, dt * 1e3))
eng.close()
# free-running decode (PREDICT): stop rule held off by the stop bias, 400 decoder steps
w = init_weights_v1(d, 1234 + 5)
w["v1/decoder/stop_token_projection/bias"] = np.array([-1e3], np.float32)
eng = EngineV1(d, w)
mem = eng.encode(ids, None)
dt, out = timed(lambda: eng.decode_free(mem, lens, max_iters=400, min_iters=10))
print("v1 free-running decode B=%d: %.1f ms for %d steps = %.1f us per decoder step" % (B, dt * 1e3, out[3], dt * 1e6 / max(out[3], 1)))
eng.close()
