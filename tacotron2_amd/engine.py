"""Thin object layer over the C-ABI: one ``Engine`` = one t2_ctx + one finalized t2_model.

The classes in modules.py / rnn_wrappers.py / helpers.py / tacotron_v2.py (mirrors of the
reference API) hold no arithmetic; they all end up here.  torch is used for device memory and
streams only.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .hparams import T2Dims
from .weights import check_weights


def _dims_struct(d: T2Dims) -> _lib.T2Dims:
    s = _lib.T2Dims()
    for f in ("num_symbols", "embedding_dim", "enc_conv_layers", "enc_kernel", "enc_units",
              "attention_lstm_units", "decoder_lstm_units", "attention_dim", "attention_filters",
              "attention_kernel", "num_mels", "outputs_per_step", "n_feed_frame", "postnet_layers",
              "postnet_kernel", "postnet_channels", "zoneout_cell", "zoneout_output", "bn_eps", "forget_bias"):
        setattr(s, f, getattr(d, f))
    s.cumulative_weights = int(d.cumulative_weights)
    if len(d.prenet_units) > _lib.T2_MAX_PRENET_LAYERS:
        raise ValueError("too many prenet layers")
    s.num_prenet_layers = len(d.prenet_units)
    for i, u in enumerate(d.prenet_units):
        s.prenet_units[i] = u
    return s


def _i64(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def _hp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def state_shapes(d: T2Dims, B: int, T_in: int):
    """Shapes of the tensors t2_state_get / t2_decode_final_state return (include/t2mi.h)."""
    A, D = d.attention_lstm_units, d.decoder_lstm_units
    return {"attention_c": (B, A), "attention_h": (B, A), "lstm1_c": (B, D), "lstm1_h": (B, D), "lstm2_c": (B, D), "lstm2_h": (B, D),
            "attention": (B, d.enc_units), "alignments": (B, T_in), "attention_state": (B, T_in), "keys": (B, T_in, d.attention_dim),
            "attention_output": (B, A), "lstm1_output": (B, D), "decoder_output": (B, D)}


class DecoderState:
    """Opaque decoder state (t2_state): the zero_state / state of the reference's cell stack."""

    def __init__(self, engine: "Engine", handle, B, T_in):
        self.engine, self._h, self.B, self.T_in = engine, handle, B, T_in
        engine._states.add(self)

    def get(self, name: str) -> torch.Tensor:
        out = torch.empty(state_shapes(self.engine.dims, self.B, self.T_in)[name], dtype=torch.float32, device=self.engine.device)
        with self.engine._ordered():
            _lib.check(self.engine.lib.t2_state_get(self._h, name.encode(), out.data_ptr()))
        return out

    def reset(self):
        with self.engine._ordered():
            _lib.check(self.engine.lib.t2_state_reset(self._h))

    def close(self):
        if self._h is not None and getattr(self.engine, "_model", None):
            self.engine.lib.t2_state_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One t2_ctx: a GPU plus the stream the library enqueues on.  The library's stream is its own
    (capturable) torch stream; every call orders it after torch's current stream on entry and torch's
    current stream after it on exit (``ordered``)."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("tacotron2_amd needs an MI355X (gfx950) GPU: there is no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(self.device)
        self.handle = C.c_void_p()
        _lib.check(self.lib.t2_ctx_create(device, C.c_void_p(self.stream.cuda_stream), C.byref(self.handle)))

    def f32(self, x) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    @contextlib.contextmanager
    def ordered(self):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        try:
            yield
        finally:
            cur.wait_stream(self.stream)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.t2_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_shared: Dict[int, Context] = {}


def shared_context(device: int = 0) -> Context:
    """A process-wide Context per GPU for calls that are not tied to a model (losses, optimizer)."""
    if device not in _shared:
        _shared[device] = Context(device)
    return _shared[device]


class Engine:
    def __init__(self, dims: T2Dims, weights: Dict[str, np.ndarray], device: int = 0):
        check_weights(dims, weights)
        self.context = Context(device)
        self.lib = self.context.lib
        self.dims = dims
        self.device = self.context.device
        self.stream = self.context.stream
        self._states = weakref.WeakSet()
        self._ctx = self.context.handle
        self._model = C.c_void_p()
        ds = _dims_struct(dims)
        _lib.check(self.lib.t2_model_create(self._ctx, C.byref(ds), C.byref(self._model)))
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self.lib.t2_model_set_weight(self._model, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        _lib.check(self.lib.t2_model_finalize(self._model))
        # tuning hook: T2MI_OPTIONS="name=value,name=value" applies library options to every engine
        for kv in filter(None, os.environ.get("T2MI_OPTIONS", "").split(",")):
            k, v = kv.split("=")
            self.set_option(k.strip(), int(v))

    # -- plumbing -------------------------------------------------------------------------
    def synchronize(self):
        _lib.check(self.lib.t2_ctx_synchronize(self._ctx))

    def set_option(self, name: str, value: int):
        _lib.check(self.lib.t2_model_set_option(self._model, name.encode(), int(value)))

    def get_stat(self, name: str) -> float:
        v = C.c_double(0)
        _lib.check(self.lib.t2_model_get_stat(self._model, name.encode(), C.byref(v)))
        return v.value

    def _f32(self, x) -> torch.Tensor:
        return self.context.f32(x)

    def _ordered(self):
        return self.context.ordered()

    def close(self):
        for st in list(getattr(self, "_states", [])):
            st.close()
        if getattr(self, "_model", None):
            self.lib.t2_model_destroy(self._model)
            self._model = None
        if getattr(self, "_ctx", None):
            self.context.close()
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hot path -------------------------------------------------------------------------
    def encode(self, ids, lens=None) -> torch.Tensor:
        ids = _i64(ids)
        lens = _i64(lens)
        B, T = ids.shape
        out = torch.empty((B, T, self.dims.enc_units), dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_encode(self._model, _hp(ids), _hp(lens), B, T, out.data_ptr()))
        return out

    def encode_embedded(self, emb, lens=None) -> torch.Tensor:
        emb = self._f32(emb)
        lens = _i64(lens)
        B, T, _ = emb.shape
        out = torch.empty((B, T, self.dims.enc_units), dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_encode_embedded(self._model, emb.data_ptr(), _hp(lens), B, T, out.data_ptr()))
        return out

    def postnet(self, mel) -> torch.Tensor:
        mel = self._f32(mel)
        B, T, M = mel.shape
        assert M == self.dims.num_mels
        out = torch.empty_like(mel)
        with self._ordered():
            _lib.check(self.lib.t2_postnet(self._model, mel.data_ptr(), B, T, out.data_ptr()))
        return out

    def postnet_training(self, mel, drop_rate: float = 0.5, seed: int = 0):
        """PostNetV2.call(is_training=True): batch-statistics BN + dropout (t2_train_postnet_forward).  Returns
        (mel_post [B, T, M], batch statistics [layers, 2, channels])."""
        mel = self._f32(mel)
        B, T, M = mel.shape
        assert M == self.dims.num_mels
        out = torch.empty_like(mel)
        stats = torch.empty((self.dims.postnet_layers, 2, self.dims.postnet_channels), dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_train_postnet_forward(self._model, mel.data_ptr(), B, T, float(drop_rate), int(seed) & 0xFFFFFFFF,
                                                         out.data_ptr(), stats.data_ptr()))
        return out, stats

    def new_state(self, memory, lens=None) -> DecoderState:
        memory = self._f32(memory)
        lens = _i64(lens)
        B, T, _ = memory.shape
        h = C.c_void_p()
        with self._ordered():
            _lib.check(self.lib.t2_state_create(self._model, memory.data_ptr(), _hp(lens), B, T, C.byref(h)))
        return DecoderState(self, h, B, T)

    def decode_step(self, state: DecoderState, prev_frames):
        d = self.dims
        prev = self._f32(prev_frames)
        assert prev.shape == (state.B, d.num_mels * d.n_feed_frame)
        mel = torch.empty((state.B, d.num_mels * d.outputs_per_step), dtype=torch.float32, device=self.device)
        stop = torch.empty((state.B, 1), dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_decode_step(self._model, state._h, prev.data_ptr(), mel.data_ptr(), stop.data_ptr()))
        return mel, stop

    def decode_step_train(self, state: DecoderState, prev_frames, prenet_drop_rate: float, sample_zoneout: bool, seed: int):
        """One cell call with sampled PreNet dropout / zoneout (t2_decode_step_train)."""
        d = self.dims
        prev = self._f32(prev_frames)
        assert prev.shape == (state.B, d.num_mels * d.n_feed_frame)
        mel = torch.empty((state.B, d.num_mels * d.outputs_per_step), dtype=torch.float32, device=self.device)
        stop = torch.empty((state.B, 1), dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_decode_step_train(self._model, state._h, prev.data_ptr(), float(prenet_drop_rate), int(bool(sample_zoneout)),
                                                     int(seed) & 0xFFFFFFFF, mel.data_ptr(), stop.data_ptr()))
        return mel, stop

    def decode_teacher(self, memory, lens, targets, teacher_forcing=True):
        d = self.dims
        memory, targets = self._f32(memory), self._f32(targets)
        lens = _i64(lens)
        B, T_in, _ = memory.shape
        T_out = targets.shape[1]
        S = T_out // d.outputs_per_step
        mel = torch.empty((B, T_out, d.num_mels), dtype=torch.float32, device=self.device)
        stop = torch.empty((B, S), dtype=torch.float32, device=self.device)
        align = torch.empty((S, B, T_in), dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_decode_teacher(self._model, memory.data_ptr(), _hp(lens), targets.data_ptr(), B, T_in, T_out,
                                              int(bool(teacher_forcing)), mel.data_ptr(), stop.data_ptr(), align.data_ptr()))
        return mel, stop.unsqueeze(-1), align

    def decode_free_buffers(self, B, T_in, max_iters):
        d = self.dims
        mel = torch.empty((B, max_iters * d.outputs_per_step, d.num_mels), dtype=torch.float32, device=self.device)
        stop = torch.empty((B, max_iters), dtype=torch.float32, device=self.device)
        align = torch.empty((max_iters, B, T_in), dtype=torch.float32, device=self.device)
        return mel, stop, align

    def decode_free(self, memory, lens, max_iters=None, min_iters=None, buffers=None):
        d = self.dims
        memory = self._f32(memory)
        lens = _i64(lens)
        B, T_in, _ = memory.shape
        max_iters = d.max_iters if max_iters is None else max_iters
        min_iters = d.min_iters if min_iters is None else min_iters
        mel, stop, align = buffers if buffers is not None else self.decode_free_buffers(B, T_in, max_iters)
        steps = C.c_int(0)
        with self._ordered():
            _lib.check(self.lib.t2_decode_free(self._model, memory.data_ptr(), _hp(lens), B, T_in, max_iters, min_iters,
                                           mel.data_ptr(), stop.data_ptr(), align.data_ptr(), C.byref(steps)))
        n = steps.value
        return mel[:, :n * d.outputs_per_step], stop[:, :n].unsqueeze(-1), align[:n], n

    def decode_final_state(self, name: str, B: int, T_in: int) -> torch.Tensor:
        """One tensor of the cell stack's state after the last decode loop (final_state of dynamic_decode)."""
        out = torch.empty(state_shapes(self.dims, B, T_in)[name], dtype=torch.float32, device=self.device)
        with self._ordered():
            _lib.check(self.lib.t2_decode_final_state(self._model, name.encode(), out.data_ptr()))
        return out

    def decode_free_launch(self, memory, lens, max_iters, min_iters, buffers, events=None):
        """events: optional (start, end) torch.cuda.Event pair recorded on the library's stream around the call
        (the state set-up before the first decoder step is inside the bracket)."""
        lens = _i64(lens)
        B, T_in, _ = memory.shape
        mel, stop, align = buffers
        with self._ordered():
            if events is not None:
                events[0].record(self.stream)
            _lib.check(self.lib.t2_decode_free_launch(self._model, memory.data_ptr(), _hp(lens), B, T_in, max_iters, min_iters,
                                                  mel.data_ptr(), stop.data_ptr(), align.data_ptr()))
            if events is not None:
                events[1].record(self.stream)

    def decode_free_finish(self) -> int:
        steps = C.c_int(0)
        with self._ordered():
            _lib.check(self.lib.t2_decode_free_finish(self._model, C.byref(steps)))
        return steps.value
