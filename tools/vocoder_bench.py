"""Griffin-Lim throughput (t2_griffin_lim, reference defaults: n_fft 2048, hop 250, win 1000, 60 iterations) and its
error against the CPU restatement: python tools/vocoder_bench.py [B] [T]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from tacotron2_amd.audio import Audio, AudioHParams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 800
hp = AudioHParams()
rng = np.random.default_rng(0)
t = np.arange(T)[:, None] / T; f = np.arange(hp.num_freq)[None, :] / hp.num_freq
spec = np.stack([(0.55 + 0.25 * np.sin(2 * np.pi * (3 * t + 5 * f + b)) * np.exp(-3 * f) + 0.1 * rng.standard_normal((T, hp.num_freq))).astype(np.float32) for b in range(B)])
audio = Audio(hp)
x = torch.tensor(spec, device="cuda")
audio.inv_spectrogram_tf(x); torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps): y = audio.inv_spectrogram_tf(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
n_fft, hop, win = audio._stft_parameters()
iters = hp.griffin_lim_iters
bytes_iter = B * T * (hp.num_freq * (8 + 8 + 4) + win * 4 * 2)       # Z write + read, S read, frames write + read (overlap reads hit L2)
flops = B * T * (2 * iters + 1) * 5 * n_fft * 11
print("B=%d T=%d: %.3f ms per call = %.1f s of audio in %.3f ms (x%.0f real time); %.1f GB/s algorithmic (%.3f of 8 TB/s), %.2f TFLOP/s of FFT butterflies; %d launches"
      % (B, T, dt * 1e3, B * y.shape[-1] / hp.sample_rate, dt * 1e3, B * y.shape[-1] / hp.sample_rate / dt,
         bytes_iter * (iters + 0.5) / dt / 1e9, bytes_iter * (iters + 0.5) / dt / 1e9 / 8000, flops / dt / 1e12, 2 * iters + 3))
if "--check" in sys.argv:
    from oracle import audio_oracle as AO          # checker only
    t0 = time.perf_counter(); want = AO.inv_spectrogram(spec[0]); t_cpu = time.perf_counter() - t0
    got = y[0].cpu().numpy()
    print("oracle: %.2f s on the CPU for one utterance; max |diff| / max |wav| = %.2e" % (t_cpu, np.abs(got - want).max() / np.abs(want).max()))
