"""Mirror of the reference's tacotron/tacotron_v1.py (EncoderV1 :18, AttentionRNNV1 :56, DecoderRNNV1 :63,
DecoderV1 :94, PostNet :156), of CBHG (tacotron/modules.py:163-232) and of the two V1 model_fns
(tacotron/models.py:15-133, :185-263) on the HIP library.  No arithmetic lives here: ``EngineV1`` binds
the ``t2v1_*`` entry points of include/t2mi.h; the classes keep the reference's constructor arguments,
call signatures, return values and error behaviour.  The block classes run the inference arithmetic
(dropout off, batch norm with moving statistics; ``DecoderV1(is_training=True)`` selects the TrainingHelper
exactly as the reference does and requires ``drop_rate == 0`` -- the reference's own test,
tests/decoder_test.py:47, runs that way); the training arithmetic (dropout, batch statistics, tapes, the
hand-written backward passes) lives behind ``PostNetTrainer``, ``EncoderV1Trainer`` and ``DecoderV1Trainer``
(``t2v1_*_train_*``), which ``PostNet(is_training=True)``, ``EncoderV1(is_training=True)`` and the TRAIN mode
of both model_fns use."""
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

    def get_stat(self, name: str) -> float:
        out = C.c_double(0.0)
        _lib.check(self.lib.t2v1_model_get_stat(self._model, name.encode(), C.byref(out)))
        return out.value

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
    """modules.py:163-232.  ``which`` selects the encoder's (0) or the PostNet's (1) instance of the bound engine.
    ``is_training=True`` (batch-statistics batch norm, modules.py:153) runs the CBHG of a trainer on the flat buffer of that
    block's variables: ``flat_params`` / ``spec`` (``training.flatten_spec`` of the embedding + v1/encoder/ variables for
    ``which=0``, of the v1/postnet/ variables for ``which=1``)."""

    def __init__(self, out_units, conv_channels, max_filter_width, projection1_out_channels, projection2_out_channels,
                 num_highway, is_training, engine: EngineV1 = None, which: int = 0, name=None, dtype=None,
                 flat_params: Optional[torch.Tensor] = None, spec=None):
        assert out_units % 2 == 0                                        # modules.py:169
        if is_training and (flat_params is None or spec is None):
            raise ValueError("CBHG(is_training=True) needs flat_params and spec: batch-statistics batch norm runs on the training step's buffer")
        self.out_units, self._engine, self._which = out_units, engine, which
        self.is_training, self.flat_params, self.spec, self.trainer = bool(is_training), flat_params, spec, None

    def __call__(self, inputs, input_lengths=None, **kwargs):
        return self.call(inputs, input_lengths=input_lengths)

    def call(self, inputs, input_lengths=None, **kwargs):
        if not self.is_training:
            return self._engine.cbhg(self._which, inputs, input_lengths)
        eng = self._engine
        x = eng._f32(inputs)
        B, T = int(x.shape[0]), int(x.shape[1])
        if self.trainer is None or (self.trainer.B, self.trainer.T) != (B, T):
            if self.trainer is not None:
                self.trainer.close()
            self.trainer = (PostNetTrainer if self._which == 1 else EncoderV1Trainer)(eng, B, T, self.spec)
        out = eng._out(B, T, self.out_units)
        lens = _i64(input_lengths)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_cbhg_train_forward(self.trainer._h, self.trainer._flat(self.flat_params, "flat_params").data_ptr(), x.data_ptr(),
                                                       _hp(lens), out.data_ptr()))
        return out

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.out_units)


class EncoderV1:
    """tacotron_v1.py:18-52.  ``is_training=True`` with ``drop_rate > 0`` (PreNet dropout, batch-statistics batch norm in the
    CBHG) runs ``EncoderV1Trainer`` and needs the flat parameter buffer it reads: ``flat_params`` / ``spec``
    (``training.flatten_spec`` of the embedding + v1/encoder/ variables); ``seed`` advances per call."""

    def __init__(self, is_training, cbhg_out_units=256, conv_channels=128, max_filter_width=16, projection1_out_channels=128,
                 projection2_out_channels=128, num_highway=4, prenet_out_units=(256, 128), drop_rate=0.5,
                 engine: EngineV1 = None, name=None, dtype=None, flat_params: Optional[torch.Tensor] = None, spec=None, seed: int = 0):
        self.prenet_out_units = tuple(prenet_out_units)
        self.cbhg_out_units = cbhg_out_units
        self._engine = engine
        self.is_training, self.drop_rate = bool(is_training), float(drop_rate)
        self.flat_params, self.spec, self.seed, self.trainer = flat_params, spec, int(seed), None
        if self.is_training and (flat_params is None or spec is None):
            raise ValueError("EncoderV1(is_training=True) needs flat_params and spec (training.flatten_spec of the embedding and "
                             "v1/encoder/ variables): batch-statistics batch norm and dropout run on the training step's buffer")

    def __call__(self, inputs, input_lengths=None, **kwargs):
        return self.call(inputs, input_lengths=input_lengths)

    def call(self, inputs, input_lengths=None, **kwargs):
        assert self.prenet_out_units[0] == inputs.shape[2]               # tacotron_v1.py:41-44
        if not self.is_training:
            return self._engine.encode_embedded(inputs, input_lengths)
        B, T = int(inputs.shape[0]), int(inputs.shape[1])
        if self.trainer is None or (self.trainer.B, self.trainer.T) != (B, T):
            if self.trainer is not None:
                self.trainer.close()
            self.trainer = EncoderV1Trainer(self._engine, B, T, self.spec)
        out = self.trainer.forward_embedded(self.flat_params, inputs, input_lengths, drop_rate=self.drop_rate, seed=self.seed)
        self.seed += 1
        return out

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
    -> [S, B, T_in] (models.py:53).  ``is_training=True`` with ``drop_rate > 0`` (PreNet dropout, ``modules.py:90``) runs the
    training forward of ``DecoderV1Trainer`` on the flat buffer of the decoder's variables: ``flat_params`` / ``spec``
    (``training.flatten_spec(DecoderV1Trainer.decoder_weights(w))``); ``seed`` advances per call; ``self.trainer`` then holds the
    tape for ``DecoderV1Trainer.backward``."""

    def __init__(self, prenet_out_units=(256, 128), drop_rate=0.5, attention_out_units=256, decoder_out_units=256, num_mels=80,
                 outputs_per_step=2, max_iters=200, n_feed_frame=1, min_iters=10, engine: EngineV1 = None, name=None, dtype=None,
                 flat_params: Optional[torch.Tensor] = None, spec=None, seed: int = 0):
        self.flat_params, self.spec, self.seed, self.trainer = flat_params, spec, int(seed), None
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
        eng, r = self._engine, self.outputs_per_step
        d = eng.dims
        if (self._prenet_out_units, self.attention_out_units, self.decoder_out_units, self.num_mels, r, self.n_feed_frame) != \
                (tuple(d.decoder_prenet_out_units), d.attention_out_units, d.decoder_out_units, d.num_mels, d.outputs_per_step, d.n_feed_frame):
            raise ValueError("DecoderV1 arguments do not match the bound engine's dimensions")
        if is_training and self._drop_rate > 0:
            if self.flat_params is None or self.spec is None:
                raise ValueError("DecoderV1(is_training=True, drop_rate > 0) needs flat_params and spec (training.flatten_spec of "
                                 "DecoderV1Trainer.decoder_weights): PreNet dropout runs on the training step's buffer")
            B, T_in = int(source.shape[0]), int(source.shape[1])
            S = min(int(target.shape[1]) // r, self.max_iters)           # maximum_iterations (:150)
            if self.trainer is None or (self.trainer.B, self.trainer.T_in, self.trainer.S) != (B, T_in, S):
                if self.trainer is not None:
                    self.trainer.close()
                self.trainer = DecoderV1Trainer(eng, B, T_in, S, self.spec)
            mel, stop, align = self.trainer.forward(self.flat_params, source, memory_sequence_length, eng._f32(target)[:, :S * r].contiguous(),
                                                    drop_rate=self._drop_rate, seed=self.seed)
            self.seed += 1
            return mel, stop, (SimpleNamespace(alignment_history=_AlignmentHistory(align)), None, None)
        if is_training or is_validation:
            tf_on = bool(is_training or teacher_forcing)                 # TrainingHelper always feeds targets (helpers.py:182)
            mel, stop, align = eng.decode_teacher(source, memory_sequence_length, target, teacher_forcing=tf_on)
            S = min(stop.shape[1], self.max_iters)                       # maximum_iterations (:150)
            mel, stop, align = mel[:, :S * r], stop[:, :S], align[:S]
        else:
            mel, stop, align, _ = eng.decode_free(source, memory_sequence_length, max_iters=self.max_iters, min_iters=self.min_iters)
        return mel, stop, (SimpleNamespace(alignment_history=_AlignmentHistory(align)), None, None)


class PostNetTrainer:
    """One t2v1_ptrainer: tape + workspaces of PostNet(is_training=True) for a fixed (B, T) and of its backward pass
    (include/t2mi.h, "Tacotron-1 PostNet training step").  The trainable state is ONE flat fp32 device buffer holding every
    ``v1/postnet/`` variable in TF layout (``training.flatten_spec`` order); the kernels read it in place, so an optimizer step
    needs no refresh.  ``export()`` rebuilds the engine's inference copies (PREDICT / EVAL) from the buffer."""

    PREFIX = "v1/postnet/"

    def __init__(self, engine: EngineV1, B: int, T: int, spec):
        self.engine, self.B, self.T, self.spec = engine, int(B), int(T), spec
        self._spec = spec
        self._h = C.c_void_p()
        _lib.check(engine.lib.t2v1_postnet_trainer_create(engine._model, self.B, self.T, C.byref(self._h)))
        for name, shape, off, size in spec:
            _lib.check(engine.lib.t2v1_postnet_trainer_bind(self._h, name.encode(), int(off)))

    @classmethod
    def postnet_weights(cls, weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        return {k: v for k, v in weights.items() if k.startswith(cls.PREFIX)}

    def _flat(self, t: torch.Tensor, what: str) -> torch.Tensor:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                and t.numel() == training.flat_size(self.spec)):
            raise ValueError(f"{what} must be a contiguous fp32 device tensor of {training.flat_size(self.spec)} elements")
        return t

    def forward(self, flat_params: torch.Tensor, mel) -> torch.Tensor:
        eng = self.engine
        mel = eng._f32(mel)
        if tuple(mel.shape) != (self.B, self.T, eng.dims.num_mels):
            raise ValueError(f"mel {tuple(mel.shape)} does not match the trainer's (B, T, num_mels) = {(self.B, self.T, eng.dims.num_mels)}")
        self._params = self._flat(flat_params, "flat_params")           # the library reads it again in backward
        out = eng._out(self.B, self.T, eng.dims.num_freq)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_postnet_train_forward(self._h, flat_params.data_ptr(), mel.data_ptr(), out.data_ptr()))
        return out

    def backward(self, spec_target, spec_loss_mask, flat_grads: torch.Tensor, loss_count: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Gradient of spec_loss(last forward, spec_target, mask) (models.py:208-211) into flat_grads; returns dev [3] =
        (sum, non-zero weight count, loss).  loss_count: dev [1], the count of a larger data-parallel batch."""
        eng = self.engine
        y, m = eng._f32(spec_target), eng._f32(spec_loss_mask)
        if tuple(y.shape) != (self.B, self.T, eng.dims.num_freq) or tuple(m.shape) != (self.B, self.T):
            raise ValueError("label shapes do not match the trainer's (B, T)")
        self._flat(flat_grads, "flat_grads")
        cnt = None if loss_count is None else eng._f32(loss_count).reshape(1)
        out = eng._out(3)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_postnet_train_backward(self._h, y.data_ptr(), m.data_ptr(), None if cnt is None else cnt.data_ptr(),
                                                           flat_grads.data_ptr(), out.data_ptr()))
        return out

    def update_moving_stats(self, flat_params: torch.Tensor, momentum: float = 0.99):
        with self.engine.context.ordered():
            _lib.check(self.engine.lib.t2v1_postnet_trainer_update_moving_stats(self._h, self._flat(flat_params, "flat_params").data_ptr(),
                                                                                float(momentum)))

    def export(self, flat_params: torch.Tensor):
        with self.engine.context.ordered():
            _lib.check(self.engine.lib.t2v1_postnet_trainer_export(self._h, self._flat(flat_params, "flat_params").data_ptr()))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.engine, "_model", None):
            self.engine.lib.t2v1_postnet_trainer_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EncoderV1Trainer(PostNetTrainer):
    """The encoder half of SingleSpeakerTacotronV1Model's training step (models.py:40-47 with is_training=True): Embedding ->
    PreNets with dropout -> CBHG with batch-statistics batch norm, forward with a tape and the backward pass from d loss /
    d memory to the embedding table (include/t2mi.h, t2v1_encoder_train_*).  Same flat-buffer convention as ``PostNetTrainer``
    over the ``embedding`` and ``v1/encoder/`` variables; ``update_moving_stats`` / ``export`` are inherited."""

    def __init__(self, engine: EngineV1, B: int, T: int, spec):
        self.engine, self.B, self.T, self.spec = engine, int(B), int(T), spec
        self._spec = spec
        self._h = C.c_void_p()
        _lib.check(engine.lib.t2v1_encoder_trainer_create(engine._model, self.B, self.T, C.byref(self._h)))
        for name, shape, off, size in spec:
            _lib.check(engine.lib.t2v1_postnet_trainer_bind(self._h, name.encode(), int(off)))

    @staticmethod
    def encoder_weights(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        return {k: v for k, v in weights.items() if k == "embedding" or k.startswith("v1/encoder/")}

    def forward(self, flat_params: torch.Tensor, ids, lens=None, drop_rate: float = 0.5, seed: int = 0) -> torch.Tensor:
        eng = self.engine
        ids, lens = _i64(ids), _i64(lens)
        if tuple(ids.shape) != (self.B, self.T):
            raise ValueError("ids do not match the trainer's (B, T_in)")
        self._params = self._flat(flat_params, "flat_params")
        out = eng._out(self.B, self.T, eng.dims.cbhg_out_units)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_encoder_train_forward(self._h, flat_params.data_ptr(), _hp(ids), _hp(lens), float(drop_rate),
                                                          int(seed) & 0xFFFFFFFF, out.data_ptr()))
        return out

    def forward_embedded(self, flat_params: torch.Tensor, emb, lens=None, drop_rate: float = 0.5, seed: int = 0) -> torch.Tensor:
        eng = self.engine
        emb, lens = eng._f32(emb), _i64(lens)
        if tuple(emb.shape) != (self.B, self.T, eng.dims.embedding_dim):
            raise ValueError("input does not match the trainer's (B, T_in, embedding_dim)")
        self._params = self._flat(flat_params, "flat_params")
        out = eng._out(self.B, self.T, eng.dims.cbhg_out_units)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_encoder_train_forward_embedded(self._h, flat_params.data_ptr(), emb.data_ptr(), _hp(lens), float(drop_rate),
                                                                   int(seed) & 0xFFFFFFFF, out.data_ptr()))
        return out

    def backward(self, dmemory, flat_grads: torch.Tensor, want_input_grad: bool = False):
        """Gradients of every encoder variable (and of the table after ``forward``) for d loss / d memory = ``dmemory``."""
        eng = self.engine
        dm = eng._f32(dmemory)
        if tuple(dm.shape) != (self.B, self.T, eng.dims.cbhg_out_units):
            raise ValueError("dmemory does not match the trainer's (B, T_in, cbhg_out_units)")
        self._flat(flat_grads, "flat_grads")
        demb = eng._out(self.B, self.T, eng.dims.embedding_dim) if want_input_grad else None
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_encoder_train_backward(self._h, dm.data_ptr(), flat_grads.data_ptr(), None if demb is None else demb.data_ptr()))
        return demb


class DecoderV1Trainer:
    """DecoderV1.call(is_training=True) with a tape and its backward pass (include/t2mi.h, t2v1_train_decoder_*): the decoder
    half of SingleSpeakerTacotronV1Model's training step, over the ``v1/decoder/`` and ``v1/attention/`` variables of one flat
    fp32 device buffer.  One trainer per (B, T_in, S)."""

    def __init__(self, engine: EngineV1, B: int, T_in: int, S: int, spec):
        self.engine, self.B, self.T_in, self.S, self.spec = engine, int(B), int(T_in), int(S), spec
        self._h = C.c_void_p()
        _lib.check(engine.lib.t2v1_decoder_trainer_create(engine._model, self.B, self.T_in, self.S, C.byref(self._h)))
        for name, shape, off, size in spec:
            _lib.check(engine.lib.t2v1_decoder_trainer_bind(self._h, name.encode(), int(off)))

    @staticmethod
    def decoder_weights(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        return {k: v for k, v in weights.items() if k.startswith(("v1/decoder/", "v1/attention/"))}

    def _flat(self, t, what):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
                and t.numel() == training.flat_size(self.spec)):
            raise ValueError(f"{what} must be a contiguous fp32 device tensor of {training.flat_size(self.spec)} elements")
        return t

    def forward(self, flat_params, memory, lens, target, drop_rate: float = 0.5, seed: int = 0):
        """-> (mel [B, S*r, M], stop [B, S, 1], alignments [S, B, T_in])."""
        eng, d = self.engine, self.engine.dims
        memory, target, lens = eng._f32(memory), eng._f32(target), _i64(lens)
        if tuple(memory.shape) != (self.B, self.T_in, d.cbhg_out_units) or tuple(target.shape) != (self.B, self.S * d.outputs_per_step, d.num_mels):
            raise ValueError("memory / target do not match the trainer's (B, T_in, S)")
        self._params = self._flat(flat_params, "flat_params")
        mel, stop, align = torch.empty_like(target), eng._out(self.B, self.S), eng._out(self.S, self.B, self.T_in)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_train_decoder_forward(self._h, flat_params.data_ptr(), memory.data_ptr(), _hp(lens), target.data_ptr(), float(drop_rate),
                                                          int(seed) & 0xFFFFFFFF, mel.data_ptr(), stop.data_ptr(), align.data_ptr()))
        return mel, stop.unsqueeze(-1), align

    def set_loss_counts(self, counts: Optional[torch.Tensor]):
        self._counts = None if counts is None else self.engine._f32(counts)
        _lib.check(self.engine.lib.t2v1_decoder_trainer_set_loss_counts(self._h, None if self._counts is None else self._counts.data_ptr()))

    def backward(self, done, spec_loss_mask, binary_loss_mask, flat_grads):
        """-> (dmemory [B, T_in, E], losses dev [3] = mel, done, total)."""
        eng, d = self.engine, self.engine.dims
        done, m1, m2 = eng._f32(done), eng._f32(spec_loss_mask), eng._f32(binary_loss_mask)
        if tuple(done.shape) != (self.B, self.S) or tuple(m2.shape) != (self.B, self.S) or tuple(m1.shape) != (self.B, self.S * d.outputs_per_step):
            raise ValueError("label shapes do not match the trainer's (B, S)")
        self._flat(flat_grads, "flat_grads")
        dmem, losses = eng._out(self.B, self.T_in, d.cbhg_out_units), eng._out(3)
        with eng.context.ordered():
            _lib.check(eng.lib.t2v1_train_decoder_backward(self._h, done.data_ptr(), m1.data_ptr(), m2.data_ptr(), flat_grads.data_ptr(),
                                                           dmem.data_ptr(), losses.data_ptr()))
        return dmem, losses

    def close(self):
        if getattr(self, "_h", None) and getattr(self.engine, "_model", None):
            self.engine.lib.t2v1_decoder_trainer_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PostNet:
    """tacotron_v1.py:156-186: CBHG on the mel spectrogram, then Dense(num_freq).  ``is_training=True`` (batch-statistics batch
    norm, modules.py:149-157) needs the flat parameter buffer the training step owns: ``flat_params`` / ``spec`` (see
    ``PostNetTrainer``); the call then leaves a tape for ``trainer.backward``."""

    def __init__(self, is_training, num_freq, cbhg_out_units=256, conv_channels=128, max_filter_width=8,
                 projection1_out_channels=256, projection2_out_channels=80, num_highway=4, engine: EngineV1 = None,
                 name=None, dtype=None, flat_params: Optional[torch.Tensor] = None, spec=None):
        self.is_training, self.num_freq, self._engine = bool(is_training), num_freq, engine
        self.flat_params, self.spec, self.trainer = flat_params, spec, None
        if self.is_training and (flat_params is None or spec is None):
            raise ValueError("PostNet(is_training=True) needs flat_params and spec (training.flatten_spec of the v1/postnet/ variables)")

    def __call__(self, inputs, **kwargs):
        return self.call(inputs)

    def call(self, inputs, **kwargs):
        if not self.is_training:
            return self._engine.postnet(inputs)
        B, T = int(inputs.shape[0]), int(inputs.shape[1])
        if self.trainer is None or (self.trainer.B, self.trainer.T) != (B, T):
            if self.trainer is not None:
                self.trainer.close()
            self.trainer = PostNetTrainer(self._engine, B, T, self.spec)
        return self.trainer.forward(self.flat_params, inputs)

    def compute_output_shape(self, input_shape):
        return (input_shape[0], input_shape[1], self.num_freq)


class ModeKeys:
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"


class SingleSpeakerTacotronV1Model:
    """models.py:15-133.  PREDICT returns the reference's prediction dict (:122-129); EVAL the losses of the free-feeding and
    the teacher-forced pass (:94-107); TRAIN (:57-76) runs one whole training step on the model's flat parameter buffer
    (``self.flat_params``: embedding, v1/encoder/, v1/attention/, v1/decoder/; created from ``weights`` at the first TRAIN call):
    EncoderV1 / DecoderV1 in training mode (PreNet dropout, batch-statistics batch norm; the encoder without lengths, :47),
    spec_loss + binary_loss, backward through decoder and encoder, clip_by_global_norm(1.0) + Adam at the (decayed) rate, the
    batch norms' UPDATE_OPS.  The next EVAL / PREDICT call rebuilds the engine from the trained variables."""

    def __init__(self, params: V1Dims, weights: Optional[Dict[str, np.ndarray]] = None, device: int = 0, seed: int = 1234,
                 engine: EngineV1 = None):
        self.params = params
        self._weights = weights if weights is not None else (None if engine is not None else init_weights_v1(params, seed))
        self._device, self.seed = device, seed
        self._own_engine = engine is None
        self.engine = engine if engine is not None else EngineV1(params, self._weights, device)
        self._make_decoder()
        self.flat_params = self.flat_grads = self.optimizer = None
        self._enc_tr = self._dec_tr = None
        self.global_step, self._stale = 0, False

    def _make_decoder(self):
        p = self.params
        self.decoder = DecoderV1(prenet_out_units=p.decoder_prenet_out_units, drop_rate=0.0, attention_out_units=p.attention_out_units,
                                 decoder_out_units=p.decoder_out_units, num_mels=p.num_mels, outputs_per_step=p.outputs_per_step,
                                 max_iters=p.max_iters, n_feed_frame=p.n_feed_frame, min_iters=p.min_iters, engine=self.engine)

    @staticmethod
    def _hp(params, name, default):
        if params is None:
            return default
        return params.get(name, default) if isinstance(params, dict) else getattr(params, name, default)

    def _setup_training(self, params):
        if self._weights is None:
            raise ValueError("SingleSpeakerTacotronV1Model(weights=...) is needed for mode TRAIN (the initial values of the variables)")
        tw = {k: v for k, v in self._weights.items() if not k.startswith("v1/postnet/")}
        self.spec = training.flatten_spec(tw)
        self.flat_params = torch.from_numpy(training.flatten(tw, self.spec)).to(self.engine.device)
        self.flat_grads = torch.zeros_like(self.flat_params)
        init = self._hp(params, "initial_learning_rate", 0.002)
        lr = (lambda step: training.learning_rate_decay(init, step)) if self._hp(params, "decay_learning_rate", True) else init
        self._adam = (lr, self._hp(params, "adam_beta1", 0.9), self._hp(params, "adam_beta2", 0.999), self._hp(params, "adam_eps", 1e-8))
        self.optimizer = training.AdamOptimizer(self.engine, self.flat_params.numel(), *self._adam)

    def _trainers(self, B, T_in, S):
        if self._enc_tr is None or (self._enc_tr.B, self._enc_tr.T) != (B, T_in):
            if self._enc_tr is not None:
                self._enc_tr.close()
            self._enc_tr = EncoderV1Trainer(self.engine, B, T_in, self.spec)
        if self._dec_tr is None or (self._dec_tr.B, self._dec_tr.T_in, self._dec_tr.S) != (B, T_in, S):
            if self._dec_tr is not None:
                self._dec_tr.close()
            self._dec_tr = DecoderV1Trainer(self.engine, B, T_in, S, self.spec)
        return self._enc_tr, self._dec_tr

    def train_step(self, features, labels, params=None, group=None):
        """One TRAIN step (models.py:40-76).  Data parallel over ``group``: as ``training.Trainer.train_step`` (global non-zero
        loss-weight counts first, summed gradients, rank-folded mask seed, moving statistics averaged)."""
        if self.flat_params is None:
            self._setup_training(params)
        f = features if isinstance(features, dict) else features.__dict__
        lab = labels if isinstance(labels, dict) else labels.__dict__
        eng, d = self.engine, self.params
        ids = _i64(f["source"])
        B, T_in = ids.shape
        S = int(np.shape(lab["mel"])[1]) // d.outputs_per_step
        enc_tr, dec_tr = self._trainers(B, T_in, S)
        world = training._world(group)
        seed = training.rank_seed(self.seed + self.global_step, training._rank(group), world)
        self.flat_grads.zero_()
        counts = None
        if world > 1:
            counts = training.global_loss_counts(eng._f32(lab["spec_loss_mask"]), eng._f32(lab["binary_loss_mask"]), d.num_mels, group)
        dec_tr.set_loss_counts(counts)
        memory = enc_tr.forward(self.flat_params, ids, None, drop_rate=self._hp(params, "encoder_prenet_drop_rate", 0.5), seed=seed)
        mel, stop, align = dec_tr.forward(self.flat_params, memory, f["source_length"], lab["mel"],
                                          drop_rate=self._hp(params, "decoder_prenet_drop_rate", 0.5), seed=seed)
        dmem, losses = dec_tr.backward(lab["done"], lab["spec_loss_mask"], lab["binary_loss_mask"], self.flat_grads)
        enc_tr.backward(dmem, self.flat_grads)
        if world > 1:
            training.allreduce_sum(self.flat_grads, group)
            m1, m2 = eng._f32(lab["spec_loss_mask"]), eng._f32(lab["binary_loss_mask"])
            own = torch.stack([(m1 != 0).sum().to(torch.float32) * float(d.num_mels), (m2 != 0).sum().to(torch.float32)])
            num = training.allreduce_sum(losses[:2] * own, group)
            ml, dl = num[0] / counts[0].clamp(min=1), num[1] / counts[1].clamp(min=1)
            losses = torch.stack([ml, dl, ml + dl])
        lr = self.optimizer.learning_rate(self.global_step) if callable(self.optimizer.learning_rate) else self.optimizer.learning_rate
        self.global_step = self.optimizer.apply_gradients(self.flat_grads, self.flat_params, global_step=self.global_step, clip_norm=1.0)
        enc_tr.update_moving_stats(self.flat_params)
        if world > 1:
            training.Trainer._average_moving_stats(enc_tr, self.flat_params, group, world)
        self._stale = True
        return {"loss": losses[2], "mel_loss": losses[0], "done_loss": losses[1], "global_step": self.global_step, "learning_rate": lr,
                "mel": mel, "alignment": align.permute(1, 2, 0)}

    def trained_weights(self) -> Dict[str, np.ndarray]:
        out = dict(self._weights)
        out.update(training.unflatten(self.flat_params, self.spec))
        return out

    def _refresh_engine(self):
        """EVAL / PREDICT after TRAIN: the inference decoder's packed operands have no device-side refresh, so the engine is
        rebuilt from the trained variables (a host round trip, once per switch from training to inference)."""
        if not self._own_engine:
            raise RuntimeError("the engine was passed in: build a new EngineV1 from model.trained_weights()")
        self._weights = self.trained_weights()
        for t in (self._enc_tr, self._dec_tr):
            if t is not None:
                t.close()
        self._enc_tr = self._dec_tr = None
        if self.optimizer is not None:
            state = self.optimizer.state()
            self.optimizer.close()
        self.engine.close()
        self.engine = EngineV1(self.params, self._weights, self._device)
        if self.optimizer is not None:
            self.optimizer = training.AdamOptimizer(self.engine, self.flat_params.numel(), *self._adam)
            self.optimizer.load_state(*state)
        self._make_decoder()
        self._stale = False

    def model_fn(self, features, labels=None, mode=ModeKeys.PREDICT, params=None):
        if mode == ModeKeys.TRAIN:
            return self.train_step(features, labels, params)
        if self._stale:
            self._refresh_engine()
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
        for t in (self._enc_tr, self._dec_tr):
            if t is not None:
                t.close()
        if self.optimizer is not None:
            self.optimizer.close()
        self.engine.close()


class TacotronV1PostNetModel:
    """models.py:185-263.  ``audio``: a ``tacotron2_amd.audio.Audio`` (the reference's constructor argument, :185-187);
    when given, PREDICT adds the Griffin-Lim waveform of the predicted spectrogram (:256).

    TRAIN (:213-234): ``model_fn`` runs one whole training step -- PostNet(is_training=True), L1 spec_loss, backward,
    clip_by_global_norm(1.0) + Adam at the (decayed) learning rate, the batch-norm UPDATE_OPS -- on the model's flat parameter
    buffer (``self.flat_params``, created from ``weights`` at the first TRAIN call) and returns {"loss", "global_step"}.
    The engine's inference copies are refreshed lazily, at the next EVAL / PREDICT call."""

    def __init__(self, params: V1Dims, engine: EngineV1, audio=None, weights: Optional[Dict[str, np.ndarray]] = None):
        self.params, self.engine, self.audio = params, engine, audio
        self.post_net = self._make_post_net(False)
        self._weights = weights
        self.flat_params = self.flat_grads = self.optimizer = self._train_net = None
        self.global_step, self._stale = 0, False

    def _make_post_net(self, is_training, **kw):
        p = self.params
        return PostNet(is_training, p.num_freq, p.post_net_cbhg_out_units, p.post_net_conv_channels, p.post_net_max_filter_width,
                       p.post_net_projection1_out_channels, p.post_net_projection2_out_channels, p.post_net_num_highway,
                       engine=self.engine, **kw)

    @staticmethod
    def _hp(params, name, default):
        if params is None:
            return default
        return params.get(name, default) if isinstance(params, dict) else getattr(params, name, default)

    def _setup_training(self, params):
        if self._weights is None:
            raise ValueError("TacotronV1PostNetModel(weights=...) is needed for mode TRAIN (the initial values of the variables)")
        w = PostNetTrainer.postnet_weights(self._weights)
        self.spec = training.flatten_spec(w)
        self.flat_params = torch.from_numpy(training.flatten(w, self.spec)).to(self.engine.device)
        self.flat_grads = torch.zeros_like(self.flat_params)
        init = self._hp(params, "initial_learning_rate", 0.002)
        lr = (lambda step: training.learning_rate_decay(init, step)) if self._hp(params, "decay_learning_rate", True) else init
        self.optimizer = training.AdamOptimizer(self.engine, self.flat_params.numel(), lr, self._hp(params, "adam_beta1", 0.9),
                                                self._hp(params, "adam_beta2", 0.999), self._hp(params, "adam_eps", 1e-8))
        self._train_net = self._make_post_net(True, flat_params=self.flat_params, spec=self.spec)

    def train_step(self, features, labels, params=None, group=None):
        """One TRAIN step (models.py:204-234).  Data parallel (``group`` of world > 1): as ``training.Trainer.train_step`` --
        the ranks sum their non-zero loss-weight counts first, each differentiates sum_rank / count_total, gradients are summed
        by one all-reduce (RCCL), the moving statistics are averaged."""
        if self.flat_params is None:
            self._setup_training(params)
        f = features if isinstance(features, dict) else features.__dict__
        lab = labels if isinstance(labels, dict) else labels.__dict__
        eng, net = self.engine, self._train_net
        world = training._world(group)
        self.flat_grads.zero_()
        linear_output = net(f["mel"])
        count = None
        mask = eng._f32(lab["spec_loss_mask"])
        if world > 1:
            count = (mask != 0).sum().to(torch.float32).reshape(1) * float(self.params.num_freq)
            training.allreduce_sum(count, group)
        parts = net.trainer.backward(lab["spec"], mask, self.flat_grads, count)
        loss = parts[2]
        if world > 1:
            training.allreduce_sum(self.flat_grads, group)
            loss = training.allreduce_sum(parts[:1].clone(), group)[0] / count[0].clamp(min=1)
        self.global_step = self.optimizer.apply_gradients(self.flat_grads, self.flat_params, global_step=self.global_step, clip_norm=1.0)
        net.trainer.update_moving_stats(self.flat_params)
        if world > 1:
            training.Trainer._average_moving_stats(net.trainer, self.flat_params, group, world)
        self._stale = True
        return {"loss": loss, "global_step": self.global_step, "spec": linear_output}

    def trained_weights(self) -> Dict[str, np.ndarray]:
        return training.unflatten(self.flat_params, self.spec)

    def model_fn(self, features, labels=None, mode=ModeKeys.PREDICT, params=None):
        if mode == ModeKeys.TRAIN:
            return self.train_step(features, labels, params)
        if self._stale:                                                  # predict with the trained variables
            self._train_net.trainer.export(self.flat_params)
            self._stale = False
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
