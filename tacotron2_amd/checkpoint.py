"""Weight containers on disk and import of TensorFlow-1 variables (SURVEY.md section 8f rank 2, section 5
"Checkpoint / resume").

The reference keeps its weights in ``tf.estimator`` checkpoints (``models.py:17,131-133``).  TensorFlow cannot run in
this pipeline, so the exchange format is a flat ``.npz`` of named fp32 arrays in TF layouts: what
``{n: reader.get_tensor(n) for n, _ in tf.train.list_variables(ckpt)}`` gives on a machine that has TF, saved with
``numpy.savez``.  Files are read with ``allow_pickle=False`` only.

``import_tf_variables`` maps TF variable names to the names of ``weights.weight_shapes``.  The reference never wires
the V2 blocks into a model (README.md:93), so the scope PREFIXES of a V2 checkpoint are whatever the assembling code
chose; the importer therefore keys on what the reference does fix -- ``embedding`` (modules.py:59), ``conv1d_{i}``
(tacotron_v2.py:49,189,194), ``attention_variable`` / ``attention_bias`` (:69,73), ``location_features_convolution`` /
``location_features_layer`` (:110,114) -- on TF's own layer names (``memory_layer``, ``query_layer``, ``fw`` / ``bw``,
``dense``, ``batch_normalization``, ``kernel``, ``bias``) and on shapes, and takes an explicit ``name_map`` for the rest.
Optimizer slots (``Adam``, ``beta1_power`` ...) and ``global_step`` are ignored.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping, Optional

import numpy as np

from .hparams import T2Dims
from .weights import check_weights, weight_shapes

_IGNORED = re.compile(r"(^|/)(Adam(_\d+)?|beta[12]_power|global_step|ExponentialMovingAverage)($|/)")
_BN_LEAVES = ("gamma", "beta", "moving_mean", "moving_variance")
_ATTENTION = {"memory_layer": "attention/memory_layer/kernel", "query_layer": "attention/query_layer/kernel",
              "location_features_layer": "attention/location_features_layer/kernel"}


def save_npz(path: str, weights: Mapping[str, np.ndarray]) -> None:
    """One array per variable, fp32, names as in ``weights.weight_shapes`` (or any other flat naming)."""
    np.savez(path, **{k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()})


def load_npz(path: str, d: Optional[T2Dims] = None) -> Dict[str, np.ndarray]:
    """Read a container written by ``save_npz``; with ``d`` the names and shapes are checked against the model."""
    with np.load(path, allow_pickle=False) as z:
        w = {k: np.asarray(z[k], dtype=np.float32) for k in z.files}
    if d is not None:
        check_weights(d, w)
    return w


def _trailing_index(token: str) -> int:
    m = re.search(r"_(\d+)$", token)
    return int(m.group(1)) if m else 0


def import_tf_variables(variables: Mapping[str, np.ndarray], d: T2Dims,
                        name_map: Optional[Mapping[str, str]] = None) -> Dict[str, np.ndarray]:
    """TF-1 variable name -> array  ==>  the weight dict ``Engine`` / ``t2_model_set_weight`` take.

    ``name_map``: explicit ``{tf_name: our_name}`` entries, applied first.  Raises ``KeyError`` when a model variable
    finds no source, ``ValueError`` when two sources claim one variable or a shape does not fit.
    """
    shapes = weight_shapes(d)
    out: Dict[str, np.ndarray] = {}
    src_of: Dict[str, str] = {}
    E, A, D = d.enc_units, d.attention_lstm_units, d.decoder_lstm_units
    M, r = d.num_mels, d.outputs_per_step
    lstm_rows = {"decoder/attention_lstm": d.prenet_units[-1] + E + A, "decoder/lstm_1": A + E + D, "decoder/lstm_2": D + D}

    def put(ours: str, tf_name: str, arr: np.ndarray):
        if ours not in shapes:
            raise KeyError(f"{tf_name}: no model variable named {ours}")
        if ours in src_of:
            raise ValueError(f"{ours} claimed by both {src_of[ours]} and {tf_name}; pass name_map")
        if tuple(arr.shape) != tuple(shapes[ours]):
            raise ValueError(f"{tf_name} -> {ours}: shape {tuple(arr.shape)} != {tuple(shapes[ours])}")
        out[ours] = np.ascontiguousarray(arr, dtype=np.float32)
        src_of[ours] = tf_name

    def classify(name: str, arr: np.ndarray) -> Optional[str]:
        tok = name.split("/")
        leaf = tok[-1]
        low = name.lower()
        if leaf == "embedding":
            return "embedding"
        if leaf in ("attention_variable", "attention_bias"):
            return "attention/" + leaf
        for t in tok:
            if t in _ATTENTION and leaf == "kernel":
                return _ATTENTION[t]
            if t == "location_features_convolution" and leaf in ("kernel", "bias"):
                return "attention/location_features_convolution/" + leaf
        conv = [t for t in tok if re.fullmatch(r"conv1d_\d+", t)]
        if conv:                                                   # Conv1d blocks of EncoderV2 / PostNetV2
            block = "postnet" if "post" in low else "encoder"
            if leaf == "kernel":
                return f"{block}/{conv[0]}/kernel"
            if leaf in _BN_LEAVES:
                return f"{block}/{conv[0]}/batch_normalization/{leaf}"
            return None
        if "fw" in tok or "bw" in tok:                             # bidirectional_dynamic_rnn scopes
            if leaf in ("kernel", "bias"):
                return f"encoder/bidirectional_rnn/{'fw' if 'fw' in tok else 'bw'}/lstm_cell/{leaf}"
            return None
        pre = [t for t in tok if re.fullmatch(r"pre_?net(_\d+)?", t)]
        if pre and leaf in ("kernel", "bias"):                     # PreNet layers: pre_net, pre_net_1, ...
            return f"decoder/prenet_{_trailing_index(pre[0])}/dense/{leaf}"
        if leaf == "kernel" and arr.ndim == 2 and arr.shape[1] in (4 * A, 4 * D):
            hits = [k for k, rows in lstm_rows.items() if arr.shape == (rows, 4 * (A if k.endswith("attention_lstm") else D))]
            if len(hits) == 1:
                return hits[0] + "/kernel"
            cells = [t for t in tok if re.fullmatch(r"cell_\d+", t)]          # MultiRNNCell scopes cell_0 / cell_1 / cell_2
            if cells:
                return ("decoder/attention_lstm", "decoder/lstm_1", "decoder/lstm_2")[_trailing_index(cells[0])] + "/kernel"
            return None
        if leaf == "bias" and arr.ndim == 1 and arr.shape[0] in (4 * A, 4 * D) and "dense" not in low:
            cells = [t for t in tok if re.fullmatch(r"cell_\d+", t)]
            if cells:
                return ("decoder/attention_lstm", "decoder/lstm_1", "decoder/lstm_2")[_trailing_index(cells[0])] + "/bias"
            if A != D:
                return ("decoder/attention_lstm/bias" if arr.shape[0] == 4 * A else None)
            return None
        if leaf == "kernel" and arr.ndim == 2:                     # the three Dense layers after the cells
            if "post" in low and arr.shape == (d.postnet_channels, M):
                return "postnet/projection/kernel"
            if arr.shape == (D, M * r):
                return "decoder/out_projection/kernel"
            if arr.shape == (D, 1):
                return "decoder/stop_token_projection/kernel"
        if leaf == "bias" and arr.ndim == 1:
            if "post" in low and arr.shape == (M,):
                return "postnet/projection/bias"
            if arr.shape == (M * r,) and (r > 1 or "post" not in low):
                return "decoder/out_projection/bias"
            if arr.shape == (1,):
                return "decoder/stop_token_projection/bias"
        return None

    name_map = dict(name_map or {})
    for tf_name, arr in variables.items():
        tf_name = tf_name[:-2] if tf_name.endswith(":0") else tf_name
        arr = np.asarray(arr)
        if tf_name in name_map:
            put(name_map[tf_name], tf_name, arr)
            continue
        if _IGNORED.search(tf_name):
            continue
        ours = classify(tf_name, arr)
        if ours is not None:
            put(ours, tf_name, arr)
    missing = [k for k in shapes if k not in out]
    if missing:
        raise KeyError("no TensorFlow variable found for: " + ", ".join(missing) + " (pass name_map)")
    check_weights(d, out)
    return out


def tf_style_names(d: T2Dims) -> Dict[str, str]:
    """our name -> the TF-1 variable name a V2 model assembled like the V1 model_fn (models.py:18-61) would produce with
    default layer names.  Used by the tests and as documentation of what the importer expects."""
    m: Dict[str, str] = {"embedding": "embedding/embedding"}
    for block, scope, n, first in (("encoder", "encoder_v2", d.enc_conv_layers, 0), ("postnet", "post_net_v2", d.postnet_layers, 1)):
        for i in range(first, first + n):
            m[f"{block}/conv1d_{i}/kernel"] = f"{scope}/conv1d_{i}/conv1d/kernel"
            for v in _BN_LEAVES:
                m[f"{block}/conv1d_{i}/batch_normalization/{v}"] = f"{scope}/conv1d_{i}/batch_normalization/{v}"
    for direction in ("fw", "bw"):
        for leaf in ("kernel", "bias"):
            m[f"encoder/bidirectional_rnn/{direction}/lstm_cell/{leaf}"] = \
                f"encoder_v2/bidirectional_rnn/{direction}/zoneout_lstm_cell/lstm_cell/{leaf}"
    att = "decoder_v2/decoder/output_and_stop_token_wrapper/decoder_rnnv2/multi_rnn_cell/cell_0/attention_rnn/attention_wrapper"
    m["attention/memory_layer/kernel"] = "decoder_v2/memory_layer/kernel"
    m["attention/query_layer/kernel"] = att + "/location_sensitive_attention/query_layer/kernel"
    m["attention/attention_variable"] = att + "/location_sensitive_attention/attention_variable"
    m["attention/attention_bias"] = att + "/location_sensitive_attention/attention_bias"
    m["attention/location_features_convolution/kernel"] = att + "/location_sensitive_attention/location_features_convolution/kernel"
    m["attention/location_features_convolution/bias"] = att + "/location_sensitive_attention/location_features_convolution/bias"
    m["attention/location_features_layer/kernel"] = att + "/location_sensitive_attention/location_features_layer/kernel"
    for i in range(len(d.prenet_units)):
        scope = "pre_net" if i == 0 else f"pre_net_{i}"
        for leaf in ("kernel", "bias"):
            m[f"decoder/prenet_{i}/dense/{leaf}"] = f"{att}/decoder_pre_net_wrapper/{scope}/dense/{leaf}"
    cell = "decoder_v2/decoder/output_and_stop_token_wrapper/decoder_rnnv2/multi_rnn_cell"
    for i, ours in enumerate(("decoder/attention_lstm", "decoder/lstm_1", "decoder/lstm_2")):
        for leaf in ("kernel", "bias"):
            m[f"{ours}/{leaf}"] = f"{cell}/cell_{i}/zoneout_lstm_cell/lstm_cell/{leaf}"
    for leaf in ("kernel", "bias"):
        m[f"decoder/out_projection/{leaf}"] = f"decoder_v2/decoder/output_and_stop_token_wrapper/dense/{leaf}"
        m[f"decoder/stop_token_projection/{leaf}"] = f"decoder_v2/decoder/output_and_stop_token_wrapper/dense_1/{leaf}"
        m[f"postnet/projection/{leaf}"] = f"post_net_v2/dense/{leaf}"
    return m
