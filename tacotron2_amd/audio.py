"""Mirror of the waveform side of the reference's ``util/audio.py`` ``Audio`` class over the C-ABI
(``t2_griffin_lim``).  Feature extraction (librosa STFT / mel basis, ``load_wav``,
``spectrogram``, ``melspectrogram``: offline preprocessing) is out of scope.  No arithmetic lives here."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .engine import Context, shared_context


class AudioHParams:
    """The hparams ``Audio`` reads (reference hparams.py:15-20,88-89) with the reference's defaults."""

    def __init__(self, num_freq=1025, sample_rate=20000, frame_length_ms=50, frame_shift_ms=12.5, min_level_db=-100,
                 ref_level_db=20, griffin_lim_iters=60, power=1.5):
        self.num_freq, self.sample_rate = num_freq, sample_rate
        self.frame_length_ms, self.frame_shift_ms = frame_length_ms, frame_shift_ms
        self.min_level_db, self.ref_level_db = min_level_db, ref_level_db
        self.griffin_lim_iters, self.power = griffin_lim_iters, power


class Audio:
    def __init__(self, hparams=None, context: Context = None):
        self.hparams = hparams if hparams is not None else AudioHParams()
        self._ctx = context

    @property
    def context(self) -> Context:
        if self._ctx is None:
            self._ctx = shared_context(0)
        return self._ctx

    def _stft_parameters(self):
        """util/audio.py:96-100."""
        hp = self.hparams
        n_fft = (hp.num_freq - 1) * 2
        hop_length = int(hp.frame_shift_ms / 1000 * hp.sample_rate)
        win_length = int(hp.frame_length_ms / 1000 * hp.sample_rate)
        return n_fft, hop_length, win_length

    def inv_spectrogram_tf(self, spectrogram) -> torch.Tensor:
        """util/audio.py:55-59: normalised spectrogram [T, num_freq] (or a batch [B, T, num_freq]) -> waveform
        [(T - 1) * hop + win] (or [B, ...]) on the GPU."""
        hp = self.hparams
        ctx = self.context
        x = ctx.f32(spectrogram)
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        if x.dim() != 3 or x.shape[-1] != hp.num_freq:
            raise ValueError(f"spectrogram must be [T, {hp.num_freq}] or [B, T, {hp.num_freq}], got {tuple(x.shape)}")
        x = x.contiguous()
        B, T, F = x.shape
        _, hop, win = self._stft_parameters()
        wav = torch.empty((B, (T - 1) * hop + win), dtype=torch.float32, device=ctx.device)
        with ctx.ordered():
            _lib.check(ctx.lib.t2_griffin_lim(ctx.handle, x.data_ptr(), B, T, F, win, hop, float(hp.min_level_db),
                                              float(hp.ref_level_db), float(hp.power), int(hp.griffin_lim_iters), wav.data_ptr()))
        return wav[0] if squeeze else wav

    inv_spectrogram = inv_spectrogram_tf

    def save_wav(self, wav, path):
        """util/audio.py:43-45 (16-bit PCM, peak-normalised)."""
        from scipy.io import wavfile
        w = wav.detach().cpu().numpy() if isinstance(wav, torch.Tensor) else np.asarray(wav)
        w = w * 32767 / max(0.01, float(np.max(np.abs(w))))
        wavfile.write(path, self.hparams.sample_rate, w.astype(np.int16))
