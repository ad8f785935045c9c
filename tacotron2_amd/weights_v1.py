"""Named fp32 weights of the Tacotron v1 path (BASELINE.json configs[4]) in TensorFlow layouts.

GRU variables follow tf.nn.rnn_cell.GRUCell: ``gates/kernel [in + h, 2h]`` (column blocks r | u,
``gates/bias`` initialised to 1.0), ``candidate/kernel [in + h, h]`` (rows ``[inputs; r * h]``),
``candidate/bias`` 0.  HighwayNet: ``H`` (relu) and ``T`` (sigmoid, bias initialised to -1.0,
modules.py:107) Dense layers.  CBHG scope names are the reference's (modules.py:179,188,195):
``conv1d_K{k}``, ``proj1``, ``proj2``.  The adjustment Dense exists only when
``projection2_out_channels != out_units // 2`` (modules.py:216-217)."""
from __future__ import annotations

from typing import Dict

import numpy as np

from .hparams import V1Dims
from .weights import _glorot_uniform


def _conv_bn(s, p, k, cin, cout):
    s[f"{p}/kernel"] = (k, cin, cout)
    for v in ("gamma", "beta", "moving_mean", "moving_variance"):
        s[f"{p}/batch_normalization/{v}"] = (cout,)


def _gru(s, p, cin, h):
    s[f"{p}/gates/kernel"] = (cin + h, 2 * h)
    s[f"{p}/gates/bias"] = (2 * h,)
    s[f"{p}/candidate/kernel"] = (cin + h, h)
    s[f"{p}/candidate/bias"] = (h,)


def cbhg_shapes(s, p, cin, out_units, conv_channels, max_filter_width, proj1, proj2, num_highway):
    for k in range(1, max_filter_width + 1):
        _conv_bn(s, f"{p}/conv1d_K{k}", k, cin, conv_channels)
    _conv_bn(s, f"{p}/proj1", 3, conv_channels * max_filter_width, proj1)
    _conv_bn(s, f"{p}/proj2", 3, proj1, proj2)
    half = out_units // 2
    if proj2 != half:
        s[f"{p}/adjustment/kernel"] = (proj2, half)
        s[f"{p}/adjustment/bias"] = (half,)
    for i in range(1, num_highway + 1):
        for g in ("H", "T"):
            s[f"{p}/highway_{i}/{g}/kernel"] = (half, half)
            s[f"{p}/highway_{i}/{g}/bias"] = (half,)
    for direction in ("fw", "bw"):
        _gru(s, f"{p}/bidirectional_rnn/{direction}/gru_cell", half, half)


def weight_shapes_v1(d: V1Dims) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    s["embedding"] = (d.num_symbols, d.embedding_dim)
    cin = d.embedding_dim
    for i, u in enumerate(d.encoder_prenet_out_units):
        s[f"v1/encoder/prenet_{i}/dense/kernel"] = (cin, u)
        s[f"v1/encoder/prenet_{i}/dense/bias"] = (u,)
        cin = u
    cbhg_shapes(s, "v1/encoder/cbhg", cin, d.cbhg_out_units, d.conv_channels, d.max_filter_width,
                d.projection1_out_channels, d.projection2_out_channels, d.num_highway)
    E, A, D = d.cbhg_out_units, d.attention_out_units, d.decoder_out_units
    s["v1/attention/memory_layer/kernel"] = (E, A)
    s["v1/attention/query_layer/kernel"] = (A, A)
    s["v1/attention/attention_v"] = (A,)
    pin = d.num_mels * d.n_feed_frame
    for i, u in enumerate(d.decoder_prenet_out_units):
        s[f"v1/decoder/prenet_{i}/dense/kernel"] = (pin, u)
        s[f"v1/decoder/prenet_{i}/dense/bias"] = (u,)
        pin = u
    _gru(s, "v1/decoder/attention_gru", pin + E, A)
    s["v1/decoder/output_projection/kernel"] = (A + E, D)          # OutputProjectionWrapper, rnn_wrappers.py:174-177
    s["v1/decoder/output_projection/bias"] = (D,)
    _gru(s, "v1/decoder/gru_1", D, D)
    _gru(s, "v1/decoder/gru_2", D, D)
    s["v1/decoder/out_projection/kernel"] = (D, d.num_mels * d.outputs_per_step)
    s["v1/decoder/out_projection/bias"] = (d.num_mels * d.outputs_per_step,)
    s["v1/decoder/stop_token_projection/kernel"] = (D, 1)
    s["v1/decoder/stop_token_projection/bias"] = (1,)
    cbhg_shapes(s, "v1/postnet/cbhg", d.num_mels, d.post_net_cbhg_out_units, d.post_net_conv_channels,
                d.post_net_max_filter_width, d.post_net_projection1_out_channels, d.post_net_projection2_out_channels,
                d.post_net_num_highway)
    s["v1/postnet/linear/kernel"] = (d.post_net_cbhg_out_units, d.num_freq)
    s["v1/postnet/linear/bias"] = (d.num_freq,)
    return s


def init_weights_v1(d: V1Dims, seed: int = 1234, perturb: bool = False) -> Dict[str, np.ndarray]:
    """Fresh-graph initialisation (Glorot-uniform kernels, zero biases except GRU gates 1.0 and highway
    T -1.0, BN identity); ``perturb`` randomises biases and BN statistics for tests."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for name, shape in weight_shapes_v1(d).items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf in ("kernel", "embedding", "attention_v"):
            w[name] = _glorot_uniform(rng, shape)
        elif leaf in ("gamma", "moving_variance"):
            w[name] = (rng.uniform(0.5, 1.5, size=shape) if perturb else np.ones(shape)).astype(np.float32)
        else:
            base = 1.0 if name.endswith("gates/bias") else (-1.0 if name.endswith("/T/bias") else 0.0)
            w[name] = (base + (rng.uniform(-0.3, 0.3, size=shape) if perturb else np.zeros(shape))).astype(np.float32)
    return w


def check_weights_v1(d: V1Dims, w: Dict[str, np.ndarray]) -> None:
    for name, shape in weight_shapes_v1(d).items():
        if name not in w:
            raise KeyError(f"missing weight {name}")
        if tuple(w[name].shape) != tuple(shape):
            raise ValueError(f"weight {name}: shape {tuple(w[name].shape)} != {tuple(shape)}")
