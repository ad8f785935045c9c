"""The CPU restatement of the reference's waveform step (oracle/audio_oracle.py) against independent statements:
scipy.signal's STFT (framing, periodic Hann, zero padding at the end, FFT convention), an explicit DFT-matrix inverse,
hand-computed amplitude values, and the property Griffin-Lim exists for.  Parity with TensorFlow itself is unpinned
(TensorFlow cannot run here); these pin the restatement's conventions."""
import numpy as np
import scipy.signal

from oracle import audio_oracle as AO


def test_stft_parameters_and_window():
    assert AO.stft_parameters() == (2048, 250, 1000)                 # hparams.py:15-18 through util/audio.py:96-100
    for n in (16, 1000):
        np.testing.assert_allclose(AO.hann_periodic(n), scipy.signal.get_window("hann", n, fftbins=True), atol=1e-7)


def test_stft_against_scipy():
    rng = np.random.default_rng(0)
    win, hop, n_fft = 100, 25, 256
    y = rng.standard_normal(win + 17 * hop + 9).astype(np.float32)  # 9 trailing samples fall off (pad_end=False)
    got = AO.stft(y, win, hop, n_fft)
    w = scipy.signal.get_window("hann", win, fftbins=True)
    _, _, z = scipy.signal.stft(y.astype(np.float64), window=w, nperseg=win, noverlap=win - hop, nfft=n_fft, boundary=None,
                                padded=False, return_onesided=True)
    want = (z * w.sum()).T                                             # scipy scales by 1 / sum(window)
    assert got.shape == want.shape == (18, n_fft // 2 + 1)
    np.testing.assert_allclose(got, want, atol=2e-5)


def test_inverse_stft_against_explicit_dft():
    rng = np.random.default_rng(1)
    win, hop, n_fft, T = 12, 4, 16, 5
    F = n_fft // 2 + 1
    Z = (rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))).astype(np.complex64)
    got = AO.inverse_stft(Z, win, hop, n_fft)
    # explicit Hermitian-extended inverse DFT (imaginary parts of the DC and Nyquist bins do not contribute)
    n = np.arange(n_fft)
    want = np.zeros((T - 1) * hop + win)
    for t in range(T):
        full = np.zeros(n_fft, np.complex128)
        full[:F] = Z[t]
        full[F:] = np.conj(Z[t, 1:F - 1][::-1])
        full[0] = full[0].real
        full[F - 1] = full[F - 1].real
        x = np.array([(full * np.exp(2j * np.pi * np.arange(n_fft) * k / n_fft)).sum().real / n_fft for k in n])
        want[t * hop:t * hop + win] += x[:win] * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / win))
    np.testing.assert_allclose(got, want, atol=1e-6)


def test_inv_amp_known_values():
    # clip -> de-normalise with min_level_db = -100 -> + ref_level_db = 20 -> 10 ** (dB / 20)   (util/audio.py:51-53,108-115)
    got = AO.inv_amp(np.array([0.0, 1.0, 0.8, -3.0, 7.0], np.float32))
    np.testing.assert_allclose(got, [1e-4, 10.0, 1.0, 1e-4, 10.0], rtol=1e-5)


def test_griffin_lim_reduces_the_spectral_error():
    rng = np.random.default_rng(2)
    win, hop, n_fft = 64, 16, 128
    y0 = np.sin(2 * np.pi * 0.05 * np.arange(win + 30 * hop)) + 0.3 * rng.standard_normal(win + 30 * hop)
    S = np.abs(AO.stft(y0.astype(np.float32), win, hop, n_fft)).astype(np.float32)
    assert AO.griffin_lim(S, win, hop, n_fft, 2).shape == (30 * hop + win,)
    # distance of |STFT(y)| from the consistent scaling of S decreases with the iterations
    target = lambda y: float(np.linalg.norm(np.abs(AO.stft(y, win, hop, n_fft)) / max(1e-9, np.abs(AO.stft(y, win, hop, n_fft)).max()) - S / S.max()))
    d0, d5, d30 = (target(AO.griffin_lim(S, win, hop, n_fft, k)) for k in (0, 5, 30))
    assert d5 < d0 and d30 <= d5 * 1.01
