"""Training-step math of the reference's model_fn (tacotron/models.py:57-76) on the HIP library:
the two losses, the learning-rate schedule (T3) and the optimizer block
(T4: tf.clip_by_global_norm + tf.train.AdamOptimizer) over one flat fp32 parameter buffer, plus the
data-parallel gradient exchange of section 8e.  No arithmetic happens in Python or torch here:
torch provides device memory and the RCCL all-reduce (``torch.distributed`` backend "nccl").

``Trainer`` runs the whole step on the device (training-mode forward with a tape, hand-written backward, clip + Adam, weight
refresh); ``AdamOptimizer.apply_gradients`` takes the flat gradient buffer it fills.  Difference from the reference worth
knowing: ``dynamic_decode`` caps the training decode at ``maximum_iterations=max_iters`` too (tacotron_v1.py:150) while a
``Trainer`` runs exactly the S steps of its targets -- under the reference, targets longer than max_iters * r never reach the
model (the dataset filters them, datasets/dataset.py:172-177), and ``datasets.py`` here applies the same filter."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .engine import Context


def _ctx_of(obj) -> Context:
    return obj if isinstance(obj, Context) else obj.context


def _weighted(ctx: Context, fn, *tensors, dims) -> torch.Tensor:
    out = torch.empty(3, dtype=torch.float32, device=ctx.device)
    with ctx.ordered():
        _lib.check(fn(ctx.handle, *[t.data_ptr() for t in tensors], *dims, out.data_ptr()))
    return out


def spec_loss_parts(ctx, y_hat, y, mask) -> torch.Tensor:
    """[sum |y_hat - y| * mask, non-zero weight count, their ratio] as a dev tensor (models.py:136-144)."""
    ctx = _ctx_of(ctx)
    y_hat, y, mask = ctx.f32(y_hat), ctx.f32(y), ctx.f32(mask)
    if y_hat.shape != y.shape or y_hat.dim() != 3 or tuple(mask.shape) != tuple(y.shape[:2]):
        raise ValueError(f"spec_loss shapes: y_hat {tuple(y_hat.shape)}, y {tuple(y.shape)}, mask {tuple(mask.shape)}")
    B, T, M = y.shape
    return _weighted(ctx, ctx.lib.t2_spec_loss, y_hat, y, mask, dims=(B, T, M))


def binary_loss_parts(ctx, done_hat, done, mask) -> torch.Tensor:
    """Same triple for the stop-token loss (models.py:146-148); done_hat [B,S,1] or [B,S] logits."""
    ctx = _ctx_of(ctx)
    done_hat = ctx.f32(done_hat)
    if done_hat.dim() == 3:
        if done_hat.shape[-1] != 1:
            raise ValueError("done_hat must be [B, S, 1]")      # tf.squeeze(axis=-1) fails the same way
        done_hat = done_hat.squeeze(-1).contiguous()
    done, mask = ctx.f32(done), ctx.f32(mask)
    if done_hat.shape != done.shape or done.shape != mask.shape or done.dim() != 2:
        raise ValueError(f"binary_loss shapes: {tuple(done_hat.shape)}, {tuple(done.shape)}, {tuple(mask.shape)}")
    B, S = done.shape
    return _weighted(ctx, ctx.lib.t2_binary_loss, done_hat, done, mask, dims=(B, S))


def spec_loss(ctx, y_hat, y, mask, n_priority_freq=None, priority_w=0) -> torch.Tensor:
    """SingleSpeakerTacotronV1Model.spec_loss (models.py:136-144).  The priority-frequency blend is
    never enabled by the reference's hparams (None / 0) and is rejected here."""
    if n_priority_freq is not None and priority_w > 0:
        raise NotImplementedError("priority-frequency L1 blend (models.py:139-142) is unused by the reference's hparams")
    return spec_loss_parts(ctx, y_hat, y, mask)[2]


def binary_loss(ctx, done_hat, done, mask) -> torch.Tensor:
    """binary_loss (models.py:146-148)."""
    return binary_loss_parts(ctx, done_hat, done, mask)[2]


def learning_rate_decay(init_rate, global_step):
    """models.py:150-154 (Noam schedule with 4000 warm-up steps)."""
    warmup_steps = 4000.0
    step = float(global_step + 1)
    return init_rate * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


def allreduce_loss_parts(parts: torch.Tensor, group=None) -> torch.Tensor:
    """Section 8e: SUM_BY_NONZERO_WEIGHTS is not a plain mean, so a data-parallel job sums numerator and
    count over ranks and divides afterwards -- the value a single device would get on the whole batch."""
    import torch.distributed as dist
    nc = parts[:2].clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        allreduce_sum(nc, group)
    return torch.where(nc[1] > 0, nc[0] / nc[1].clamp(min=1), torch.zeros_like(nc[0]))


# ---- flat parameter buffer ---------------------------------------------------------------------
def flatten_spec(weights: Dict[str, np.ndarray]) -> List[Tuple[str, Tuple[int, ...], int, int]]:
    """(name, shape, offset, size) of every variable in the flat buffer: sorted by name (the order TF's
    checkpoint and tf.trainable_variables-by-name listing use), each start padded to 4 floats so that
    every variable is float4-aligned."""
    spec, off = [], 0
    for name in sorted(weights):
        shape = tuple(int(x) for x in np.shape(weights[name]))
        size = int(np.prod(shape)) if shape else 1
        spec.append((name, shape, off, size))
        off += (size + 3) & ~3
    return spec


def decoder_prefix_end(spec) -> int:
    """End offset of the attention/ + decoder/ variables in the flat buffer.  Names are sorted, so these come first and form
    one contiguous prefix: the part of the gradient that is final when the decoder backward has run (its all-reduce is
    started before the encoder backward, ``Trainer.train_step``)."""
    end = 0
    for name, shape, off, size in spec:
        if not (name.startswith("attention/") or name.startswith("decoder/")):
            break
        assert 0 <= off - end < 4, "attention/ and decoder/ variables must form a contiguous prefix of the flat buffer"
        end = off + size
    return end


def flat_size(spec) -> int:
    name, shape, off, size = spec[-1]
    return off + ((size + 3) & ~3)


def flatten(weights: Dict[str, np.ndarray], spec=None) -> np.ndarray:
    spec = spec or flatten_spec(weights)
    flat = np.zeros(flat_size(spec), np.float32)
    for name, shape, off, size in spec:
        flat[off:off + size] = np.asarray(weights[name], np.float32).reshape(-1)
    return flat


def unflatten(flat, spec) -> Dict[str, np.ndarray]:
    flat = flat.detach().cpu().numpy() if isinstance(flat, torch.Tensor) else np.asarray(flat)
    return {name: flat[off:off + size].reshape(shape).copy() for name, shape, off, size in spec}


def _world(group=None) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None) -> int:
    import torch.distributed as dist
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def global_loss_counts(spec_loss_mask, binary_loss_mask, num_mels: int, group=None) -> torch.Tensor:
    """[count_nonzero(spec weights broadcast over num_mels), count_nonzero(stop weights)] of the WHOLE data-parallel batch
    (sum all-reduce over ranks): the denominators of tf.losses' SUM_BY_NONZERO_WEIGHTS for a single-device batch of all
    ranks' rows (models.py:136-148).  Works on CPU (gloo) and device (RCCL) tensors."""
    import torch.distributed as dist
    m1, m2 = torch.as_tensor(spec_loss_mask), torch.as_tensor(binary_loss_mask)
    cnt = torch.stack([(m1 != 0).sum().to(torch.float32) * float(num_mels), (m2 != 0).sum().to(torch.float32)])
    if _world(group) > 1:
        allreduce_sum(cnt, group)
    return cnt


class _DoneWork:
    def wait(self):
        return True


def _via_host(t: torch.Tensor, group) -> bool:
    """Device tensors under the gloo backend (rehearsals of the N > 1 path on a box without RCCL peers, e.g. several ranks
    sharing one GPU) are reduced through a host copy; "nccl" (= RCCL) reduces them in place."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_sum_async(t: torch.Tensor, group=None):
    """Start a SUM all-reduce of ``t`` (a contiguous view) and return its work handle; ``wait()`` orders the current stream
    after it (backend "nccl" = RCCL: the collective runs on the communicator's stream, beside compute)."""
    import torch.distributed as dist
    if _via_host(t, group):
        allreduce_sum(t, group)
        return _DoneWork()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    import torch.distributed as dist
    if _world(group) > 1:
        if _via_host(t, group):
            c = t.detach().cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_gradients(flat_grads: torch.Tensor, group=None) -> float:
    """One sum all-reduce of the whole flat gradient buffer (RCCL over xGMI when the backend is "nccl";
    ~151 MB at profile P: a single large collective, section 8e).  Returns the factor 1 / world_size
    the optimizer multiplies in (``grad_scale``), so the mean costs no extra pass over the buffer."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        allreduce_sum(flat_grads, group)
    return 1.0 / world


class AdamOptimizer:
    """tf.train.AdamOptimizer as model_fn uses it (models.py:67-76), fused with tf.clip_by_global_norm,
    over one flat parameter buffer.  ``learning_rate`` may be a float or a callable of global_step."""

    def __init__(self, ctx, n_params: int, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.ctx = _ctx_of(ctx)
        self.n = int(n_params)
        self.learning_rate = learning_rate
        self._h = C.c_void_p()
        _lib.check(self.ctx.lib.t2_optim_create(self.ctx.handle, self.n, float(beta1), float(beta2), float(epsilon),
                                                C.byref(self._h)))

    def _check(self, t: torch.Tensor, what: str) -> torch.Tensor:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.n):
            raise ValueError(f"{what} must be a contiguous fp32 device tensor of {self.n} elements")
        return t

    def global_norm(self, grads: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
        """tf.global_norm of grad_scale * grads (dev scalar)."""
        self._check(grads, "grads")
        out = torch.empty(1, dtype=torch.float32, device=self.ctx.device)
        with self.ctx.ordered():
            _lib.check(self.ctx.lib.t2_optim_global_norm(self._h, grads.data_ptr(), float(grad_scale), out.data_ptr()))
        return out[0]

    def apply_gradients(self, grads: torch.Tensor, params: torch.Tensor, global_step: int = 0, clip_norm: float = 1.0,
                        grad_scale: float = 1.0) -> int:
        """clip_by_global_norm(grads, clip_norm) then Adam on ``params`` in place; returns global_step + 1
        (apply_gradients increments the global step, models.py:76)."""
        self._check(grads, "grads")
        self._check(params, "params")
        lr = self.learning_rate(global_step) if callable(self.learning_rate) else self.learning_rate
        with self.ctx.ordered():
            _lib.check(self.ctx.lib.t2_optim_step(self._h, params.data_ptr(), grads.data_ptr(), float(lr), float(clip_norm),
                                                  float(grad_scale)))
        return global_step + 1

    def state(self):
        m = torch.empty(self.n, dtype=torch.float32, device=self.ctx.device)
        v = torch.empty_like(m)
        step = C.c_int64(0)
        with self.ctx.ordered():
            _lib.check(self.ctx.lib.t2_optim_get_state(self._h, m.data_ptr(), v.data_ptr(), C.byref(step)))
        return m, v, step.value

    def load_state(self, m: torch.Tensor, v: torch.Tensor, step: int):
        self._check(m, "m")
        self._check(v, "v")
        with self.ctx.ordered():
            _lib.check(self.ctx.lib.t2_optim_set_state(self._h, m.data_ptr(), v.data_ptr(), int(step)))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "handle", None):
            self.ctx.lib.t2_optim_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rank_seed(seed: int, rank: int, world: int) -> int:
    """The mask seed of one data-parallel rank: ``seed`` itself on a single device, a 32-bit mix of (seed, rank) otherwise
    (the same value feeds the forward kernels and the backward kernels that regenerate the masks)."""
    if world <= 1:
        return int(seed) & 0xFFFFFFFF
    x = (int(seed) * 0x9E3779B1 + (int(rank) + 1) * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    return x


class Trainer:
    """One t2_trainer: the tape of a (B, T_in, S) teacher-forced training forward of an ``Engine``'s model and the
    workspaces of its backward pass (include/t2mi.h, "training step")."""

    def __init__(self, engine, B: int, T_in: int, S: int):
        self.engine, self.B, self.T_in, self.S = engine, int(B), int(T_in), int(S)
        self._h = C.c_void_p()
        _lib.check(engine.lib.t2_trainer_create(engine._model, self.B, self.T_in, self.S, C.byref(self._h)))

    def decoder_forward(self, memory, lens, targets, prenet_drop_rate: float = 0.5, seed: int = 0):
        """DecoderV1.call(is_training=True) with the V2 cells: (mel [B, S*r, M], stop [B, S, 1], alignments [S, B, T_in])."""
        eng, d = self.engine, self.engine.dims
        memory, targets = eng._f32(memory), eng._f32(targets)
        lens = None if lens is None else np.ascontiguousarray(np.asarray(lens), dtype=np.int64)
        if tuple(memory.shape[:2]) != (self.B, self.T_in) or tuple(targets.shape) != (self.B, self.S * d.outputs_per_step, d.num_mels):
            raise ValueError("memory / targets do not match the trainer's (B, T_in, S)")
        mel = torch.empty_like(targets)
        stop = torch.empty((self.B, self.S), dtype=torch.float32, device=eng.device)
        align = torch.empty((self.S, self.B, self.T_in), dtype=torch.float32, device=eng.device)
        with eng._ordered():
            _lib.check(eng.lib.t2_train_decoder_forward(self._h, memory.data_ptr(), None if lens is None else lens.ctypes.data_as(C.c_void_p),
                                                        targets.data_ptr(), float(prenet_drop_rate), int(seed) & 0xFFFFFFFF,
                                                        mel.data_ptr(), stop.data_ptr(), align.data_ptr()))
        return mel, stop.unsqueeze(-1), align

    def encoder_forward(self, ids, lens, drop_rate: float = 0.5, seed: int = 0):
        """Embedding + EncoderV2 in training mode: (memory [B, T_in, E], batch-norm statistics [layers, 2, E] = mean | variance)."""
        eng, d = self.engine, self.engine.dims
        ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int64)
        lens = None if lens is None else np.ascontiguousarray(np.asarray(lens), dtype=np.int64)
        if ids.shape != (self.B, self.T_in):
            raise ValueError("ids do not match the trainer's (B, T_in)")
        memory = torch.empty((self.B, self.T_in, d.enc_units), dtype=torch.float32, device=eng.device)
        stats = torch.empty((max(d.enc_conv_layers, 1), 2, d.enc_units), dtype=torch.float32, device=eng.device)
        with eng._ordered():
            _lib.check(eng.lib.t2_train_encoder_forward(self._h, ids.ctypes.data_as(C.c_void_p),
                                                        None if lens is None else lens.ctypes.data_as(C.c_void_p), float(drop_rate),
                                                        int(seed) & 0xFFFFFFFF, memory.data_ptr(), stats.data_ptr()))
        return memory, stats[:d.enc_conv_layers]

    def encoder_forward_embedded(self, emb, lens, drop_rate: float = 0.5, seed: int = 0):
        """EncoderV2.call(is_training=True) on an embedded input [B, T_in, embedding_dim] (tacotron_v2.py:53-63)."""
        eng, d = self.engine, self.engine.dims
        emb = eng._f32(emb)
        lens = None if lens is None else np.ascontiguousarray(np.asarray(lens), dtype=np.int64)
        if tuple(emb.shape) != (self.B, self.T_in, d.embedding_dim):
            raise ValueError("input does not match the trainer's (B, T_in, embedding_dim)")
        memory = torch.empty((self.B, self.T_in, d.enc_units), dtype=torch.float32, device=eng.device)
        stats = torch.empty((max(d.enc_conv_layers, 1), 2, d.enc_units), dtype=torch.float32, device=eng.device)
        with eng._ordered():
            _lib.check(eng.lib.t2_train_encoder_forward_embedded(self._h, emb.data_ptr(), None if lens is None else lens.ctypes.data_as(C.c_void_p),
                                                                 float(drop_rate), int(seed) & 0xFFFFFFFF, memory.data_ptr(), stats.data_ptr()))
        return memory, stats[:d.enc_conv_layers]

    def bind_gradients(self, spec):
        """spec: training.flatten_spec(weights) -- where each variable's gradient goes in the flat buffer."""
        self._spec = spec
        for name, shape, off, size in spec:
            _lib.check(self.engine.lib.t2_trainer_bind_gradient(self._h, name.encode(), int(off)))

    def decoder_backward(self, done, spec_loss_mask, binary_loss_mask, flat_grads: torch.Tensor):
        """BPTT of the last decoder_forward through loss = spec_loss(mel) + binary_loss(stop) (models.py:58-61).
        Fills the decoder/ and attention/ slices of flat_grads; returns (dmemory [B, T_in, E], losses dev [3] = mel, done, total)."""
        eng, d = self.engine, self.engine.dims
        done, m1, m2 = eng._f32(done), eng._f32(spec_loss_mask), eng._f32(binary_loss_mask)
        if tuple(done.shape) != (self.B, self.S) or tuple(m2.shape) != (self.B, self.S) or tuple(m1.shape) != (self.B, self.S * d.outputs_per_step):
            raise ValueError("label shapes do not match the trainer's (B, S)")
        if not (flat_grads.is_cuda and flat_grads.dtype == torch.float32 and flat_grads.is_contiguous()):
            raise ValueError("flat_grads must be a contiguous fp32 device tensor")
        dmem = torch.empty((self.B, self.T_in, d.enc_units), dtype=torch.float32, device=eng.device)
        losses = torch.empty(3, dtype=torch.float32, device=eng.device)
        with eng._ordered():
            _lib.check(eng.lib.t2_train_decoder_backward(self._h, done.data_ptr(), m1.data_ptr(), m2.data_ptr(), flat_grads.data_ptr(),
                                                         dmem.data_ptr(), losses.data_ptr()))
        return dmem, losses

    def encoder_backward(self, dmemory: torch.Tensor, flat_grads: torch.Tensor):
        """Backward of the last encoder_forward: fills the embedding and encoder/ slices of flat_grads."""
        eng = self.engine
        dmemory = eng._f32(dmemory)
        with eng._ordered():
            _lib.check(eng.lib.t2_train_encoder_backward(self._h, dmemory.data_ptr(), flat_grads.data_ptr()))

    def load_params(self, flat_params: torch.Tensor):
        """Rebuild every device copy of the weights from the flat parameter buffer (after the optimizer step)."""
        # the library keeps the POINTER: a trainer of the same model whose operands lag behind (another batch shape, EncoderV2's
        # internal trainers) re-reads this buffer at its next backward pass -- keep the tensor alive with the model
        self.engine._last_flat_params = flat_params
        with self.engine._ordered():
            _lib.check(self.engine.lib.t2_trainer_load_params(self._h, flat_params.data_ptr()))

    def update_moving_stats(self, flat_params: torch.Tensor, momentum: float = 0.99):
        """tf.layers.batch_normalization's UPDATE_OPS (models.py:75) for the encoder convs, in place in flat_params."""
        with self.engine._ordered():
            _lib.check(self.engine.lib.t2_trainer_update_moving_stats(self._h, flat_params.data_ptr(), float(momentum)))

    def set_loss_counts(self, counts: Optional[torch.Tensor]):
        """counts dev [2] from ``global_loss_counts`` (or None: this rank's own): normalisation of the next backward passes."""
        self._counts = None if counts is None else self.engine._f32(counts)            # kept alive: the library holds the pointer
        with self.engine._ordered():
            _lib.check(self.engine.lib.t2_trainer_set_loss_counts(self._h, None if self._counts is None else self._counts.data_ptr()))

    def train_step(self, optimizer, flat_params, flat_grads, features, labels, global_step=0, seed=0, prenet_drop_rate=0.5,
                   encoder_drop_rate=0.5, clip_norm=1.0, group=None):
        """model_fn(mode=TRAIN) of the reference (models.py:44-76) for the V2 blocks: training-mode forward, loss = spec_loss +
        binary_loss, backward, gradient all-reduce over the data-parallel group (RCCL), clip_by_global_norm + Adam, moving
        statistics, weights refreshed on the device.  Returns (global_step + 1, losses dev [3] = mel, done, total).

        Data parallel (world > 1): the ranks first sum their non-zero loss-weight counts; every rank's backward then
        differentiates sum_rank / count_total, the gradients are SUMMED (one all-reduce of the flat buffer, no 1/world) and the
        reported losses are sum of numerators / sum of counts -- N ranks x B rows give the gradient and the loss of one batch
        of N x B rows under SUM_BY_NONZERO_WEIGHTS.  What stays per replica, as SURVEY.md 8e says: batch-norm batch statistics;
        the moving statistics are averaged over ranks so that every rank keeps identical parameters.  The dropout / zoneout
        masks come from a counter-based generator keyed on (seed, stream, step, LOCAL row, column): every rank folds its rank
        into the seed (``rank_seed``), otherwise row i of every rank would draw the same masks and the N x B rows would carry
        only B independent noise patterns."""
        d = self.engine.dims
        world = _world(group)
        seed = rank_seed(seed, _rank(group), world)
        flat_grads.zero_()
        counts = None
        if world > 1:
            counts = global_loss_counts(self.engine._f32(labels["spec_loss_mask"]), self.engine._f32(labels["binary_loss_mask"]), d.num_mels, group)
        self.set_loss_counts(counts)
        memory, _ = self.encoder_forward(features["source"], features["source_length"], drop_rate=encoder_drop_rate, seed=seed)
        self.decoder_forward(memory, features["source_length"], labels["mel"], prenet_drop_rate=prenet_drop_rate, seed=seed)
        dmem, losses = self.decoder_backward(labels["done"], labels["spec_loss_mask"], labels["binary_loss_mask"], flat_grads)
        # The flat buffer is sorted by variable name: attention/ and decoder/ gradients (27 M of the 32 M trained parameters) are
        # one contiguous prefix and FINAL once the decoder backward has run -- their all-reduce starts now and runs beside the
        # encoder backward (async collective on the communicator's own stream); the encoder / embedding suffix follows.
        works = []
        if world > 1:
            cut = decoder_prefix_end(self._spec)
            works.append(allreduce_sum_async(flat_grads[:cut], group))
        self.encoder_backward(dmem, flat_grads)
        if world > 1:
            works.append(allreduce_sum_async(flat_grads[cut:], group))           # gradients of sum_rank / count_total: plain sum
            for wk in works:
                wk.wait()
            m1, m2 = self.engine._f32(labels["spec_loss_mask"]), self.engine._f32(labels["binary_loss_mask"])
            own = torch.stack([(m1 != 0).sum().to(torch.float32) * float(d.num_mels), (m2 != 0).sum().to(torch.float32)])
            num = allreduce_sum(losses[:2] * own, group)                         # numerators: local ratio x local count
            mel_loss, done_loss = num[0] / counts[0].clamp(min=1), num[1] / counts[1].clamp(min=1)
            losses = torch.stack([mel_loss, done_loss, mel_loss + done_loss])
        step = optimizer.apply_gradients(flat_grads, flat_params, global_step=global_step, clip_norm=clip_norm, grad_scale=1.0)
        self.update_moving_stats(flat_params)
        if world > 1:
            self._average_moving_stats(flat_params, group, world)
        self.load_params(flat_params)
        return step, losses

    def _average_moving_stats(self, flat_params, group, world):
        """Every rank updated the batch-norm moving statistics with its own batch statistics: average them so that the
        parameter buffers stay identical on all ranks (one small all-reduce)."""
        sl = [(off, size) for name, shape, off, size in self._spec if name.endswith("/moving_mean") or name.endswith("/moving_variance")]
        if not sl:
            return
        packed = torch.cat([flat_params[o:o + n] for o, n in sl])
        allreduce_sum(packed, group)
        packed /= float(world)
        pos = 0
        for o, n in sl:
            flat_params[o:o + n] = packed[pos:pos + n]
            pos += n

    def close(self):
        if getattr(self, "_h", None) and getattr(self.engine, "_model", None):
            self.engine.lib.t2_trainer_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
