"""CPU restatement of the reference's spectrogram -> waveform step (TEST INFRASTRUCTURE ONLY: imported by tests/ and
nothing else; the product path is tacotron2_amd/csrc/vocoder.h behind t2_griffin_lim).

Follows /root/reference/util/audio.py: ``inv_spectrogram_tf`` :55-59, ``inv_amp_tf`` :51-53, ``_denormalize_tf`` :114-115,
``_db_to_amp_tf`` :108-109, ``_griffin_lim_tf`` :66-82, ``_stft_tf`` / ``_istft_tf`` :88-94, ``_stft_parameters`` :96-100,
with the TensorFlow-1.x semantics of ``tf.contrib.signal.stft`` / ``inverse_stft`` restated from knowledge of TF r1.11-1.15
[TF-sem] (TensorFlow is absent here: parity unpinned at the TF boundary, as for the rest of the oracle):

* ``stft(signals, frame_length, frame_step, fft_length, window_fn=hann_window(periodic=True), pad_end=False)``: frames
  ``1 + (L - frame_length) // frame_step``; each frame times the window, zero-padded at the END to ``fft_length``, rfft.
* ``inverse_stft(stfts, frame_length, frame_step, fft_length, window_fn=hann_window(periodic=True))``: irfft to
  ``fft_length`` samples, the first ``frame_length`` kept, times the window (the reference passes no
  ``inverse_stft_window_fn``, so the synthesis window is the plain Hann window, not its overlap-add inverse), then
  ``overlap_and_add(frame_step)``: length ``(frames - 1) * frame_step + frame_length``.

Arithmetic is fp64 inside each transform and rounded to fp32 where TensorFlow holds a float32 / complex64 tensor.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def stft_parameters(num_freq=1025, sample_rate=20000, frame_shift_ms=12.5, frame_length_ms=50):
    """util/audio.py:96-100 with hparams.py:15-18 defaults -> (n_fft, hop_length, win_length) = (2048, 250, 1000)."""
    n_fft = (num_freq - 1) * 2
    return n_fft, int(frame_shift_ms / 1000 * sample_rate), int(frame_length_ms / 1000 * sample_rate)


def hann_periodic(n):
    """tf.contrib.signal.hann_window(n, periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)).astype(F32)


def stft(y, win_length, hop_length, n_fft):
    y = np.asarray(y, F32)
    n_frames = 1 + (y.shape[-1] - win_length) // hop_length
    idx = np.arange(win_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    frames = (y[idx] * hann_periodic(win_length)).astype(F32)
    return np.fft.rfft(frames.astype(np.float64), n=n_fft, axis=-1).astype(np.complex64)


def inverse_stft(stfts, win_length, hop_length, n_fft):
    stfts = np.asarray(stfts, np.complex64)
    frames = np.fft.irfft(stfts.astype(np.complex128), n=n_fft, axis=-1)[:, :win_length].astype(F32)
    frames = (frames * hann_periodic(win_length)).astype(F32)
    n_frames = frames.shape[0]
    y = np.zeros((n_frames - 1) * hop_length + win_length, np.float64)
    for t in range(n_frames):
        y[t * hop_length:t * hop_length + win_length] += frames[t]
    return y.astype(F32)


def inv_amp(spectrogram, min_level_db=-100.0, ref_level_db=20.0):
    """util/audio.py:51-53: 10 ** ((clip(S, 0, 1) * -min_level_db + min_level_db + ref_level_db) * 0.05)."""
    s = np.clip(np.asarray(spectrogram, F32), 0.0, 1.0) * F32(-min_level_db) + F32(min_level_db)
    return np.power(F32(10.0), (s + F32(ref_level_db)) * F32(0.05)).astype(F32)


def griffin_lim(S, win_length, hop_length, n_fft, iters=60):
    """util/audio.py:66-82.  S [T, num_freq] magnitudes."""
    S = np.asarray(S, F32).astype(np.complex64)
    y = inverse_stft(S, win_length, hop_length, n_fft)
    for _ in range(iters):
        est = stft(y, win_length, hop_length, n_fft)
        angles = est / np.maximum(F32(1e-8), np.abs(est)).astype(np.complex64)
        y = inverse_stft(S * angles, win_length, hop_length, n_fft)
    return y


def inv_spectrogram(spectrogram, num_freq=1025, sample_rate=20000, frame_shift_ms=12.5, frame_length_ms=50,
                    min_level_db=-100.0, ref_level_db=20.0, power=1.5, griffin_lim_iters=60):
    """util/audio.py:55-59 with hparams.py:15-20,88-89 defaults."""
    n_fft, hop, win = stft_parameters(num_freq, sample_rate, frame_shift_ms, frame_length_ms)
    S = inv_amp(spectrogram, min_level_db, ref_level_db)
    return griffin_lim(np.power(S, F32(power)).astype(F32), win, hop, n_fft, griffin_lim_iters)
