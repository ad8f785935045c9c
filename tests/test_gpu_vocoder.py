"""Griffin-Lim on the GPU (t2_griffin_lim through tacotron2_amd.audio.Audio) against the CPU restatement
(oracle/audio_oracle.py) of the reference's util/audio.py.  Tolerances are relative to the largest |sample| of the
oracle's waveform: 2e-4 after a few iterations (fp32 FFTs on both sides); the 60-iteration run of the reference's default
is an iterated projection in which fp32 rounding differences can grow, so it is held to 2e-2 and to the property
Griffin-Lim is for -- the STFT magnitude of the result approaches the requested one as closely as the oracle's does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import audio_oracle as AO


def _spec(rng, T, F):
    """A smooth, speech-like normalised spectrogram in [0, 1] (plus out-of-range values for the clip)."""
    t = np.arange(T)[:, None] / T
    f = np.arange(F)[None, :] / F
    s = 0.55 + 0.25 * np.sin(2 * np.pi * (3 * t + 5 * f)) * np.exp(-3 * f) + 0.1 * rng.standard_normal((T, F))
    s[0, :3] = (-0.2, 1.3, 0.5)
    return s.astype(np.float32)


@pytest.mark.parametrize("num_freq,sr,T,iters", [(1025, 20000, 40, 3), (257, 8000, 23, 5), (1025, 20000, 12, 0)])
def test_griffin_lim_matches_oracle(num_freq, sr, T, iters):
    from tacotron2_amd.audio import Audio, AudioHParams
    rng = np.random.default_rng(T)
    hp = AudioHParams(num_freq=num_freq, sample_rate=sr, griffin_lim_iters=iters)
    spec = _spec(rng, T, num_freq)
    want = AO.inv_spectrogram(spec, num_freq=num_freq, sample_rate=sr, griffin_lim_iters=iters)
    got = Audio(hp).inv_spectrogram_tf(spec).cpu().numpy()
    assert got.shape == want.shape
    scale = float(np.abs(want).max())
    assert float(np.abs(got - want).max()) < 2e-4 * scale, (float(np.abs(got - want).max()), scale)


def test_griffin_lim_batch_rows_independent_and_default_iterations():
    from tacotron2_amd.audio import Audio, AudioHParams
    rng = np.random.default_rng(7)
    hp = AudioHParams()                                   # the reference's defaults: 2048 / 250 / 1000, 60 iterations
    T = 30
    specs = np.stack([_spec(rng, T, hp.num_freq) for _ in range(3)])
    audio = Audio(hp)
    batch = audio.inv_spectrogram_tf(specs).cpu().numpy()
    one = audio.inv_spectrogram_tf(specs[1]).cpu().numpy()
    np.testing.assert_array_equal(batch[1], one)
    want = AO.inv_spectrogram(specs[1])
    scale = float(np.abs(want).max())
    assert float(np.abs(one - want).max()) < 2e-2 * scale
    # spectral convergence: || |STFT(y)| - S || / || S ||, GPU result vs oracle result
    n_fft, hop, win = AO.stft_parameters()
    S = np.power(AO.inv_amp(specs[1]), np.float32(1.5))
    conv = lambda y: float(np.linalg.norm(np.abs(AO.stft(y, win, hop, n_fft)) - S) / np.linalg.norm(S))
    assert abs(conv(one) - conv(want)) < 1e-3


def test_griffin_lim_argument_errors():
    from tacotron2_amd._lib import T2Error
    from tacotron2_amd.audio import Audio, AudioHParams
    with pytest.raises(T2Error):                          # n_fft = 2 * 1000: not a power of two
        Audio(AudioHParams(num_freq=1001)).inv_spectrogram_tf(np.zeros((4, 1001), np.float32))
    with pytest.raises(ValueError):
        Audio().inv_spectrogram_tf(np.zeros((4, 7), np.float32))


def test_v1_postnet_model_predict_with_audio():
    """TacotronV1PostNetModel PREDICT (models.py:248-259): mel -> CBHG PostNet -> linear spectrogram -> Griffin-Lim."""
    from tacotron2_amd.audio import Audio, AudioHParams
    from tacotron2_amd.hparams import V1_TINY
    from tacotron2_amd.tacotron_v1 import EngineV1, ModeKeys, TacotronV1PostNetModel
    from tacotron2_amd.weights_v1 import init_weights_v1
    d = V1_TINY                                           # num_freq 9 -> n_fft 16
    eng = EngineV1(d, init_weights_v1(d, 4, perturb=True))
    hp = AudioHParams(num_freq=d.num_freq, sample_rate=320, griffin_lim_iters=4)      # win 16, hop 4
    rng = np.random.default_rng(1)
    mel = rng.uniform(0, 1, (1, 14, d.num_mels)).astype(np.float32)
    pred = TacotronV1PostNetModel(d, eng, audio=Audio(hp)).model_fn({"mel": mel}, None, ModeKeys.PREDICT)["predictions"]
    spec = pred["spec"].cpu().numpy()
    want = AO.inv_spectrogram(spec[0], num_freq=d.num_freq, sample_rate=320, griffin_lim_iters=4)
    got = pred["audio"].cpu().numpy()
    assert got.shape == (1, 13 * 4 + 16)
    assert float(np.abs(got[0] - want).max()) < 2e-4 * float(np.abs(want).max())
    assert "audio" not in TacotronV1PostNetModel(d, eng).model_fn({"mel": mel}, None, ModeKeys.PREDICT)["predictions"]
    eng.close()
