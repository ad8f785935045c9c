"""Named fp32 weight containers in TensorFlow layouts.

Layouts follow what the reference's TF variables would hold:
Dense kernel ``[in, out]``; LSTM kernel ``[in + h, 4h]`` with rows ``[inputs; h_prev]`` and
column blocks ``[i | j | f | o]``; conv kernel ``[k, in, out]``; BN vectors ``[C]``.

Variable names keep the scope names the reference fixes (``"embedding"`` modules.py:59,
``conv1d_{i}`` tacotron_v2.py:49,189,194, ``attention_variable`` / ``attention_bias`` :69,73,
``location_features_convolution`` / ``location_features_layer`` :110,114, TF's ``memory_layer``,
``query_layer``, ``kernel``, ``bias``) under a flat ``block/...`` prefix.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .hparams import T2Dims


def weight_shapes(d: T2Dims) -> Dict[str, tuple]:
    """Every variable of the V2 mel-generation path with its TF-layout shape."""
    s: Dict[str, tuple] = {}
    s["embedding"] = (d.num_symbols, d.embedding_dim)
    cin = d.embedding_dim
    for i in range(d.enc_conv_layers):
        p = f"encoder/conv1d_{i}"
        s[f"{p}/kernel"] = (d.enc_kernel, cin, d.enc_units)
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"{p}/batch_normalization/{v}"] = (d.enc_units,)
        cin = d.enc_units
    h = d.enc_units // 2
    for direction in ("fw", "bw"):
        s[f"encoder/bidirectional_rnn/{direction}/lstm_cell/kernel"] = (d.enc_units + h, 4 * h)
        s[f"encoder/bidirectional_rnn/{direction}/lstm_cell/bias"] = (4 * h,)
    E, U, A, D = d.enc_units, d.attention_dim, d.attention_lstm_units, d.decoder_lstm_units
    s["attention/memory_layer/kernel"] = (E, U)
    s["attention/query_layer/kernel"] = (A, U)
    s["attention/location_features_convolution/kernel"] = (d.attention_kernel, 1, d.attention_filters)
    s["attention/location_features_convolution/bias"] = (d.attention_filters,)
    s["attention/location_features_layer/kernel"] = (d.attention_filters, U)
    s["attention/attention_variable"] = (U,)
    s["attention/attention_bias"] = (U,)
    pin = d.num_mels * d.n_feed_frame
    for i, pu in enumerate(d.prenet_units):
        s[f"decoder/prenet_{i}/dense/kernel"] = (pin, pu)
        s[f"decoder/prenet_{i}/dense/bias"] = (pu,)
        pin = pu
    s["decoder/attention_lstm/kernel"] = (pin + E + A, 4 * A)
    s["decoder/attention_lstm/bias"] = (4 * A,)
    s["decoder/lstm_1/kernel"] = (A + E + D, 4 * D)
    s["decoder/lstm_1/bias"] = (4 * D,)
    s["decoder/lstm_2/kernel"] = (D + D, 4 * D)
    s["decoder/lstm_2/bias"] = (4 * D,)
    s["decoder/out_projection/kernel"] = (D, d.num_mels * d.outputs_per_step)
    s["decoder/out_projection/bias"] = (d.num_mels * d.outputs_per_step,)
    s["decoder/stop_token_projection/kernel"] = (D, 1)
    s["decoder/stop_token_projection/bias"] = (1,)
    cin = d.num_mels
    for i in range(1, d.postnet_layers + 1):
        p = f"postnet/conv1d_{i}"
        s[f"{p}/kernel"] = (d.postnet_kernel, cin, d.postnet_channels)
        for v in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"{p}/batch_normalization/{v}"] = (d.postnet_channels,)
        cin = d.postnet_channels
    s["postnet/projection/kernel"] = (d.postnet_channels, d.num_mels)
    s["postnet/projection/bias"] = (d.num_mels,)
    return s


def _glorot_uniform(rng, shape):
    """tf.glorot_uniform_initializer: U(-l, l), l = sqrt(6 / (fan_in + fan_out)); for conv
    kernels the receptive field multiplies both fans."""
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_weights(d: T2Dims, seed: int = 1234, perturb: bool = False) -> Dict[str, np.ndarray]:
    """Random initialisation as TF would do it for a fresh graph: Glorot-uniform kernels,
    zero biases, BN gamma=1 / beta=0 / mean=0 / var=1, xavier ``attention_variable``.

    ``perturb=True`` additionally randomises biases and BN statistics (tests use it so that no
    term of the arithmetic is trivially zero or one).
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for name, shape in weight_shapes(d).items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf in ("kernel", "embedding", "attention_variable"):
            w[name] = _glorot_uniform(rng, shape)
        elif leaf in ("gamma", "moving_variance"):
            w[name] = (rng.uniform(0.5, 1.5, size=shape) if perturb else np.ones(shape)).astype(np.float32)
        else:  # bias, beta, moving_mean, attention_bias
            w[name] = (rng.uniform(-0.3, 0.3, size=shape) if perturb else np.zeros(shape)).astype(np.float32)
    return w


SHARP_GAINS = {"attention/attention_variable": 40.0, "attention/memory_layer/kernel": 3.0, "attention/query_layer/kernel": 3.0,
               "attention/location_features_convolution/kernel": 1.5, "attention/location_features_layer/kernel": 1.5}


def sharpen_attention(w: Dict[str, np.ndarray], gains: Dict[str, float] = None) -> Dict[str, np.ndarray]:
    """A copy of ``w`` whose location-sensitive attention (tacotron_v2.py:66-78,118-141) is as peaked as a trained
    model's.  Glorot-initialised attention weights give energies of ~0.1, i.e. a near-uniform softmax over T_in
    positions (alignment max ~1/T_in): every term of the score then moves the alignments by less than any usable
    tolerance and a parity test cannot see it.  Scaling ``attention_variable`` and the memory / query / location
    kernels makes the energies span tens of units: alignments peak at 0.5..1.0, the arg-max moves from step to step
    with the cumulative location features, and dropping any one term changes the alignments by O(0.1 .. 1)
    (tests/test_oracle_sensitivity.py holds that property on the oracle).  The gains are chosen so that the recurrence
    stays well conditioned: with 3 x 3 on the location path (instead of 1.5 x 1.5) the cumulative-state feedback
    amplifies rounding noise ~3.5 x per decoder step (fp32 and fp64 runs of the ORACLE differ by 7e-4 after 7 steps),
    and no tolerance means anything; with these gains the oracle's own fp32-vs-fp64 difference stays ~1e-5 over tens
    of steps (same test file)."""
    out = dict(w)
    for name, g in (SHARP_GAINS if gains is None else gains).items():
        out[name] = (w[name] * np.float32(g)).astype(np.float32)
    return out


def check_weights(d: T2Dims, w: Dict[str, np.ndarray]) -> None:
    shapes = weight_shapes(d)
    for name, shape in shapes.items():
        if name not in w:
            raise KeyError(f"missing weight {name}")
        if tuple(w[name].shape) != tuple(shape):
            raise ValueError(f"weight {name}: shape {tuple(w[name].shape)} != {tuple(shape)}")


def count_params(d: T2Dims) -> int:
    return int(sum(int(np.prod(s)) for s in weight_shapes(d).values()))
