"""Mirror of the reference's tacotron/tacotron_v1.py (EncoderV1 :18, AttentionRNNV1 :56, DecoderRNNV1 :63,
DecoderV1 :94, PostNet :156), of CBHG (tacotron/modules.py:163-232) and of the two V1 model_fns
(tacotron/models.py:15-133, :185-263) on the HIP library.  No arithmetic lives here: ``EngineV1`` binds
the ``t2v1_*`` entry points of include/t2mi.h; the classes keep the reference's constructor arguments,
call signatures, return values and error behaviour.  Inference arithmetic only (dropout off, batch norm
with moving statistics): ``is_training=True`` selects the TrainingHelper exactly as the reference does
and requires ``drop_rate == 0`` -- the reference's own test (tests/decoder_test.py:47) runs that way."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, training
from .engine import Context, _hp, _i64
from .hparams import V1Dims
from .weights_v1 import check_weights_v1, init_weights_v1


def _dims_struct(d: V1Dims) -> _lib.T2V1Dims:
    s = _lib.T2V1Dims()
    s.num_symbols, s.embedding_dim = d.num_symbols, d.embedding_dim
    s.enc_prenet_units[0], s.enc_prenet_units[1] = d.encoder_prenet_out_units
    s.cbhg_out_units, s.conv_channels, s.max_filter_width = d.cbhg_out_units, d.conv_channels, d.max_filter_width
    s.proj1_channels, s.proj2_channels, s.num_highway = d.projection1_out_channels, d.projection2_out_channels, d.num_highway
    s.dec_prenet_units[0], s.dec_prenet_units[1] = d.decoder_prenet_out_units
    s.attention_out_units, s.decoder_out_units = d.attention_out_units, d.decoder_out_units
    s.num_mels, s.outputs_per_step, s.n_feed_frame = d.num_mels, d.outputs_per_step, d.n_feed_frame
    s.post_cbhg_out_units, s.post_conv_channels = d.post_net_cbhg_out_units, d.post_net_conv_channels
    s.post_max_filter_width, s.post_proj1_channels = d.post_net_max_filter_width, d.post_net_projection1_out_channels
    s.post_proj2_channels, s.post_num_highway = d.post_net_projection2_out_channels, d.post_net_num_highway
    s.num_freq, s.bn_eps = d.num_freq, d.bn_eps
    return s


class EngineV1:
    """One t2v1_model: all V1 weights on one GPU."""

    def __init__(self, dims: V1Dims, weights: Dict[str, np.ndarray], device: int = 0):
        if len(dims.encoder_prenet_out_units) != 2 or len(dims.decoder_prenet_out_units) != 2:
            raise ValueError("the V1 path takes exactly two PreNet layers (hparams.py:41,45)")
        check_weights_v1(dims, weights)
        self.context = Context(device)
        self.lib, self.device, self.dims = self.context.lib, self.context.device, dims
        self._model = C.c_void_p()
        ds = _dims_struct(dims)
        _lib.check(self.lib.t2v1_model_create(self.context.handle, C.byref(ds), C.byref(self._model)))
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self.lib.t2v1_model_set_weight(self._model, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        _lib.check(self.lib.t2v1_model_finalize(self._model))

    def set_option(self, name: str, value: int):
        _lib.check(self.lib.t2v1_model_set_option(self._model, name.encode(), int(value)))

    def _f32(self, x):
        return self.context.f32(x)

    def _out(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def encode(self, ids, lens=None) -> torch.Tensor:
        ids, lens = _i64(ids), _i64(lens)
        B, T = ids.shape
        out = self._out(B, T, self.dims.cbhg_out_units)
        with self.context.ordered():
            _lib.check(self.lib.t2v1_encode(self._model, _hp(ids), _hp(lens), B, T, out.data_ptr()))
        return out

    def encode_embedded(self, emb, lens=None) -> torch.Tensor:
        emb, lens = self._f32(emb), _i64(lens)
        B, T, _ = emb.shape
        out = self._out(B, T, self.dims.cbhg_out_units)
        with self.context.ordered():
            _lib.check(self.lib.t2v1_encode_embedded(self._model, emb.data_ptr(), _hp(lens), B, T, out.data_ptr()))
        return out

    def cbhg(self, which: int, x, lens=None) -> torch.Tensor:
        x, lens = self._f32(x), _i64(lens)
        B, T, _ = x.shape
        units = self.dims.cbhg_out_units if which == 0 else self.dims.post_net_cbhg_out_units
        out = self._out(B, T, units)
        with self.context.ordered():
            _lib.check(self.lib.t2v1_cbhg(self._model, which, x.data_ptr(), _hp(lens), B, T, out.data_ptr()))
        return out

    def postnet(self, mel) -> torch.Tensor:
        mel = self._f32(mel)
        B, T, _ = mel.shape
        out = self._out(B, T, self.dims.num_freq)
        with self.context.ordered():
            _lib.check(self.lib.t2v1_postnet(self._model, mel.data_ptr(), B, T, out.data_ptr()))
        return out

    def decode_teacher(self, memory, lens, targets, teacher_forcing=True):
        d = self.dims
        memory, targets, lens = self._f32(memory), self._f32(targets), _i64(lens)
        B, T_in, _ = memory.shape
        T_out = targets.shape[1]
        S = T_out // d.outputs_per_step
        mel, stop, align = self._out(B, T_out, d.num_mels), self._out(B, S), self._out(S, B, T_in)
        with self.context.ordered():
            _lib.check(self.lib.t2v1_decode_teacher(self._model, memory.data_ptr(), _hp(lens), targets.data_ptr(), B, T_in, T_out,
                                                    int(bool(teacher_forcing)), mel.data_ptr(), stop.data_ptr(), align.data_ptr()))
        return mel, stop.unsqueeze(-1), align

    def decode_free(self, memory, lens, max_iters=None, min_iters=None):
        d = self.dims
        memory, lens = self._f32(memory), _i64(lens)
        B, T_in, _ = memory.shape
        max_iters = d.max_iters if max_iters is None else max_iters
        min_iters = d.min_iters if min_iters is None else min_iters
        mel, stop = self._out(B, max_iters * d.outputs_per_step, d.num_mels), self._out(B, max_iters)
        align = self._out(max_iters, B, T_in)
        steps = C.c_int(0)
        with self.context.ordered():
            _lib.check(self.lib.t2v1_decode_free(self._model, memory.data_ptr(), _hp(lens), B, T_in, max_iters, min_iters,
                                                 mel.data_ptr(), stop.data_ptr(), align.data_ptr(), C.byref(steps)))
        n = steps.value
        return mel[:, :n * d.outputs_per_step], stop[:, :n].unsqueeze(-1), align[:n], n

    def close(self):
        if getattr(self, "_model", None):
            self.lib.t2v1_model_destroy(self._model)
            self._model = None
        if getattr(self, "context", None):
            self.context.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CBHG:
    """modules.py:163-232.  ``which`` selects the encoder's (0) or the PostNet's (1) instance of the bound engine."""

    def __init__(self, out_units, conv_channels, max_filter_width, projection1_out_channels, projection2_out_channels,
                 num_highway, is_training, engine: EngineV1 = None, which: int = 0, name=None, dtype=None):
        assert out_units % 2 == 0                                        # modules.py:169
        if is_training:
            raise NotImplementedError("training-mode batch norm (batch statistics) is not built; inference arithmetic only")
        self.out_units, self._engine, self._which = out_units, engine, which

    def __call__(self, inputs, input_lengths=None, **kwargs):
        return self.call(inputs, input_lengths=input_lengths)

    def call(self, inputs, input_lengths=None, **kwargs):
        return self._engine.cbhg(self._which, inputs, input_lengths)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.out_units)


class EncoderV1:
    """tacotron_v1.py:18-52."""

    def __init__(self, is_training, cbhg_out_units=256, conv_channels=128, max_filter_width=16, projection1_out_channels=128,
                 projection2_out_channels=128, num_highway=4, prenet_out_units=(256, 128), drop_rate=0.5,
                 engine: EngineV1 = None, name=None, dtype=None):
        if is_training and drop_rate > 0:
            raise NotImplementedError("dropout sampling (training mode) is not built; use drop_rate=0 or is_training=False")
        self.prenet_out_units = tuple(prenet_out_units)
        self.cbhg_out_units = cbhg_out_units
        self._engine = engine

    def __call__(self, inputs, input_lengths=None, **kwargs):
        return self.call(inputs, input_lengths=input_lengths)

    def call(self, inputs, input_lengths=None, **kwargs):
        assert self.prenet_out_units[0] == inputs.shape[2]               # tacotron_v1.py:41-44
        return self._engine.encode_embedded(inputs, input_lengths)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.cbhg_out_units)


class _AlignmentHistory:
    def __init__(self, stacked):
        self._stacked = stacked

    def stack(self):
        return self._stacked


class DecoderV1:
    """tacotron_v1.py:94-153: same constructor arguments, ``call`` signature, helper selection (:134-147) and return
    triple ``(mel [B, S*r, M], stop [B, S, 1], final_state)`` with ``final_state[0].alignment_history.stack()``
    -> [S, B, T_in] (models.py:53)."""

    def __init__(self, prenet_out_units=(256, 128), drop_rate=0.5, attention_out_units=256, decoder_out_units=256, num_mels=80,
                 outputs_per_step=2, max_iters=200, n_feed_frame=1, min_iters=10, engine: EngineV1 = None, name=None, dtype=None):
        if decoder_out_units < 1:
            raise ValueError("Parameter output_size must be > 0: %d." % decoder_out_units)      # rnn_wrappers.py:155-156
        self._prenet_out_units = tuple(prenet_out_units)
        self._drop_rate = drop_rate
        self.attention_out_units, self.decoder_out_units = attention_out_units, decoder_out_units
        self.num_mels, self.outputs_per_step, self.max_iters, self.n_feed_frame = num_mels, outputs_per_step, max_iters, n_feed_frame
        self.min_iters = min_iters
        self._engine = engine

    def __call__(self, source, **kw):
        return self.call(source, **kw)

    def call(self, source, is_training=None, is_validation=None, teacher_forcing=False, memory_sequence_length=None, target=None):
        assert is_training is not None                                   # tacotron_v1.py:120
        assert self.n_feed_frame <= self.outputs_per_step                # helpers.py:55,100,144
        if is_training and self._drop_rate > 0:
            raise NotImplementedError("dropout sampling (training mode) is not built; use drop_rate=0 or is_training=False")
        eng, r = self._engine, self.outputs_per_step
        d = eng.dims
        if (self._prenet_out_units, self.attention_out_units, self.decoder_out_units, self.num_mels, r, self.n_feed_frame) != \
                (tuple(d.decoder_prenet_out_units), d.attention_out_units, d.decoder_out_units, d.num_mels, d.outputs_per_step, d.n_feed_frame):
            raise ValueError("DecoderV1 arguments do not match the bound engine's dimensions")
        if is_training or is_validation:
            tf_on = bool(is_training or teacher_forcing)                 # TrainingHelper always feeds targets (helpers.py:182)
            mel, stop, align = eng.decode_teacher(source, memory_sequence_length, target, teacher_forcing=tf_on)
            S = min(stop.shape[1], self.max_iters)                       # maximum_iterations (:150)
            mel, stop, align = mel[:, :S * r], stop[:, :S], align[:S]
        else:
            mel, stop, align, _ = eng.decode_free(source, memory_sequence_length, max_iters=self.max_iters, min_iters=self.min_iters)
        return mel, stop, (SimpleNamespace(alignment_history=_AlignmentHistory(align)), None, None)


class PostNet:
    """tacotron_v1.py:156-186: CBHG on the mel spectrogram, then Dense(num_freq)."""

    def __init__(self, is_training, num_freq, cbhg_out_units=256, conv_channels=128, max_filter_width=8,
                 projection1_out_channels=256, projection2_out_channels=80, num_highway=4, engine: EngineV1 = None,
                 name=None, dtype=None):
        if is_training:
            raise NotImplementedError("training-mode batch norm (batch statistics) is not built; inference arithmetic only")
        self.num_freq, self._engine = num_freq, engine

    def __call__(self, inputs, **kwargs):
        return self.call(inputs)

    def call(self, inputs, **kwargs):
        return self._engine.postnet(inputs)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.num_freq)


class ModeKeys:
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"


class SingleSpeakerTacotronV1Model:
    """models.py:15-133.  PREDICT returns the reference's prediction dict (:122-129); EVAL the losses of the
    free-feeding and the teacher-forced pass (:94-107); TRAIN is not built (no backward pass yet)."""

    def __init__(self, params: V1Dims, weights: Optional[Dict[str, np.ndarray]] = None, device: int = 0, seed: int = 1234,
                 engine: EngineV1 = None):
        self.params = params
        self.engine = engine if engine is not None else EngineV1(params, weights if weights is not None else init_weights_v1(params, seed), device)
        p = params
        self.decoder = DecoderV1(prenet_out_units=p.decoder_prenet_out_units, drop_rate=0.0, attention_out_units=p.attention_out_units,
                                 decoder_out_units=p.decoder_out_units, num_mels=p.num_mels, outputs_per_step=p.outputs_per_step,
                                 max_iters=p.max_iters, n_feed_frame=p.n_feed_frame, min_iters=p.min_iters, engine=self.engine)

    def model_fn(self, features, labels=None, mode=ModeKeys.PREDICT, params=None):
        if mode == ModeKeys.TRAIN:
            raise NotImplementedError("the training step (gradients) is not built yet")
        is_validation = mode == ModeKeys.EVAL
        f = features if isinstance(features, dict) else features.__dict__
        lab = None if labels is None else (labels if isinstance(labels, dict) else labels.__dict__)
        target = lab["mel"] if is_validation else None
        encoder_output = self.engine.encode(f["source"], None)            # models.py:46-47: no input lengths
        mel_output, stop_token, state = self.decoder(encoder_output, is_training=False, is_validation=is_validation,
                                                     memory_sequence_length=f["source_length"], target=target)
        alignment = state[0].alignment_history.stack().permute(1, 2, 0)   # models.py:53
        if is_validation:
            eng = self.engine
            mel_loss = training.spec_loss(eng, mel_output, lab["mel"], lab["spec_loss_mask"])
            done_loss = training.binary_loss(eng, stop_token, lab["done"], lab["binary_loss_mask"])
            mel_tf, stop_tf, _ = self.decoder(encoder_output, is_training=False, is_validation=True,
                                              memory_sequence_length=f["source_length"], target=target, teacher_forcing=True)
            mel_loss_tf = training.spec_loss(eng, mel_tf, lab["mel"], lab["spec_loss_mask"])
            done_loss_tf = training.binary_loss(eng, stop_tf, lab["done"], lab["binary_loss_mask"])
            return {"loss": mel_loss + done_loss, "eval_metric_ops": {
                "mel_loss": mel_loss, "done_loss": done_loss, "loss_with_teacher": mel_loss_tf + done_loss_tf,
                "mel_loss_with_teacher": mel_loss_tf, "done_loss_with_teacher": done_loss_tf},
                "mel": mel_output, "alignment": alignment}
        return {"predictions": {"id": f.get("id"), "mel": mel_output, "alignment": alignment, "source": f["source"],
                                "text": f.get("text"), "stop_token": stop_token}}

    def predict(self, features):
        return self.model_fn(features, None, ModeKeys.PREDICT)["predictions"]

    def close(self):
        self.engine.close()


class TacotronV1PostNetModel:
    """models.py:185-263.  ``audio``: a ``tacotron2_amd.audio.Audio`` (the reference's constructor argument, :185-187);
    when given, PREDICT adds the Griffin-Lim waveform of the predicted spectrogram (:256)."""

    def __init__(self, params: V1Dims, engine: EngineV1, audio=None):
        self.params, self.engine, self.audio = params, engine, audio
        self.post_net = PostNet(False, params.num_freq, params.post_net_cbhg_out_units, params.post_net_conv_channels,
                                params.post_net_max_filter_width, params.post_net_projection1_out_channels,
                                params.post_net_projection2_out_channels, params.post_net_num_highway, engine=engine)

    def model_fn(self, features, labels=None, mode=ModeKeys.PREDICT, params=None):
        if mode == ModeKeys.TRAIN:
            raise NotImplementedError("the training step (gradients) is not built yet")
        f = features if isinstance(features, dict) else features.__dict__
        linear_output = self.post_net(f["mel"])
        if mode == ModeKeys.EVAL:
            lab = labels if isinstance(labels, dict) else labels.__dict__
            loss = training.spec_loss(self.engine, linear_output, lab["spec"], lab["spec_loss_mask"])     # models.py:208-211
            return {"loss": loss, "eval_metric_ops": {"linear_loss": loss}, "spec": linear_output}
        pred = {"id": f.get("id"), "spec": linear_output, "mel": f["mel"], "ground_truth_spec": f.get("spec"),
                "ground_truth_mel": f.get("ground_truth_mel"), "alignment": f.get("alignment"),
                "source": f.get("source"), "text": f.get("text")}
        if self.audio is not None:
            pred["audio"] = self.audio.inv_spectrogram_tf(linear_output)                                  # models.py:256
        return {"predictions": pred}
