"""Mirror of the reference's tacotron/helpers.py (the tf.contrib.seq2seq.Helper protocol):
``initialize() -> (finished, first_inputs)``, ``sample(...)`` (dummy zeros),
``next_inputs(time, outputs, state, sample_ids) -> (finished, next_inputs, state)``.

Tensors are torch tensors on the decoder's device.  These objects drive the generic
``dynamic_decode`` loop in tacotron_v2.py (one C-ABI ``t2_decode_step`` per iteration); the fused
decode loops of the library (``t2_decode_teacher`` / ``t2_decode_free``) implement the same rules
on the device and are what ``DecoderV2`` uses.
"""
from __future__ import annotations

import torch


def _go_frames(batch_size, output_dim, dtype, device):
    """helpers.py:187-188."""
    return torch.zeros((batch_size, output_dim), dtype=dtype, device=device)


class _HelperBase:
    @property
    def batch_size(self):
        return self._batch_size

    @property
    def sample_ids_shape(self):
        return torch.Size([])

    @property
    def sample_ids_dtype(self):
        return torch.int32

    def sample(self, time, outputs, state, name=None):
        # return all-zero dummy tensor (helpers.py:40-42)
        return torch.zeros((self._batch_size,), dtype=torch.int32)


class InferenceHelper(_HelperBase):
    """helpers.py:13-49: legacy helper, a row finishes when its whole output equals 0.0."""

    def __init__(self, batch_size, output_dim, r, n_feed_frame=1, dtype=None, device="cuda"):
        assert n_feed_frame <= r
        self._batch_size = batch_size
        self._output_dim = output_dim
        self.n_feed_frame = n_feed_frame
        self._dtype = dtype or torch.float32
        self._device = device
        self._end_token = torch.zeros((output_dim * r,), dtype=self._dtype, device=device)

    def initialize(self, name=None):
        return (torch.zeros((self._batch_size,), dtype=torch.bool),
                _go_frames(self._batch_size, self._output_dim * self.n_feed_frame, self._dtype, self._device))

    def next_inputs(self, time, outputs, state, sample_ids=None, name=None):
        outputs, done = outputs
        finished = torch.all(outputs == self._end_token, dim=1).cpu()
        next_inputs = outputs[:, -self._output_dim * self.n_feed_frame:]
        return (finished, next_inputs, state)


class StopTokenBasedInferenceHelper(_HelperBase):
    """helpers.py:52-94."""

    def __init__(self, batch_size, output_dim, r, n_feed_frame=1, min_iters=10, dtype=None, device="cuda"):
        assert n_feed_frame <= r
        self._batch_size = batch_size
        self._output_dim = output_dim
        self.n_feed_frame = n_feed_frame
        self.min_iters = min_iters
        self._dtype = dtype or torch.float32
        self._device = device

    def initialize(self, name=None):
        return (torch.zeros((self._batch_size,), dtype=torch.bool),
                _go_frames(self._batch_size, self._output_dim * self.n_feed_frame, self._dtype, self._device))

    def next_inputs(self, time, outputs, state, sample_ids=None, name=None):
        output, done = outputs
        finished = self.is_finished(done, time)
        next_inputs = output[:, -self._output_dim * self.n_feed_frame:]
        return (finished, next_inputs, state)

    def is_finished(self, done, time):
        termination_criteria = torch.sigmoid(done) > 0.5          # [B, 1]
        minimum_requirement = time > self.min_iters
        termination = termination_criteria & bool(minimum_requirement)
        return torch.all(termination, dim=0).cpu()                # shape [1]: batch-coupled (helpers.py:94)


class ValidationHelper(_HelperBase):
    """helpers.py:97-138."""

    def __init__(self, targets, batch_size, output_dim, r, n_feed_frame=1, teacher_forcing=False):
        assert n_feed_frame <= r
        self._batch_size = batch_size
        self._output_dim = output_dim
        self.n_feed_frame = n_feed_frame
        self.num_steps = targets.shape[1] // r
        self.teacher_forcing = teacher_forcing
        self._targets = targets.reshape(batch_size, self.num_steps, output_dim * r)

    def initialize(self, name=None):
        return (torch.zeros((self._batch_size,), dtype=torch.bool),
                _go_frames(self._batch_size, self._output_dim * self.n_feed_frame, self._targets.dtype, self._targets.device))

    def next_inputs(self, time, outputs, state, sample_ids=None, name=None):
        output, done = outputs
        finished = torch.full((self._batch_size,), bool(time + 1 >= self.num_steps))
        n = self._output_dim * self.n_feed_frame
        next_inputs = self._targets[:, time, -n:] if self.teacher_forcing else output[:, -n:]
        return (finished, next_inputs, state)


class TrainingHelper(_HelperBase):
    """helpers.py:141-184: full length for every row, padding frames are not masked (:154-156)."""

    def __init__(self, targets, output_dim, r, n_feed_frame=1):
        assert n_feed_frame <= r
        self._batch_size = targets.shape[0]
        self._output_dim = output_dim
        self.n_feed_frame = n_feed_frame
        self._targets = targets.reshape(self._batch_size, targets.shape[1] // r, output_dim * r)
        num_steps = self._targets.shape[1]
        self._lengths = torch.full((self._batch_size,), num_steps)

    def initialize(self, name=None):
        return (torch.zeros((self._batch_size,), dtype=torch.bool),
                _go_frames(self._batch_size, self._output_dim * self.n_feed_frame, self._targets.dtype, self._targets.device))

    def next_inputs(self, time, outputs, state, sample_ids=None, name=None):
        output, done = outputs
        finished = (time + 1 >= self._lengths)
        next_inputs = self._targets[:, time, -self._output_dim * self.n_feed_frame:]
        return (finished, next_inputs, state)
