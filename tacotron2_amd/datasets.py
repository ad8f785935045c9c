"""Tensor contract of the reference's input pipeline (reference datasets/dataset.py).

The tf.data pipeline, TFRecord wire format and Spark preprocessing are out of scope; what the hot path's callers need
is the *contract* between that pipeline and ``model_fn``: how a target utterance is padded to a multiple of
``outputs_per_step``, what ``done`` and the two loss masks hold, which utterances are dropped, how examples are
grouped into windows and padded into batches.  This module restates those rules on NumPy arrays, so real
(preprocessed) utterances and synthetic ones reach ``TacotronV2Model.model_fn`` / ``training.Trainer`` in the same form.
No arithmetic of the hot path lives here.
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

# defaults of the reference's hparams.py:27,63,73,75,76,87
OUTPUTS_PER_STEP = 2
BATCH_SIZE = 32
APPROX_MIN_TARGET_LENGTH = 100
BATCH_BUCKET_WIDTH = 50
BATCH_NUM_BUCKETS = 50
MAX_ITERS = 500


def prepare_target(mel: np.ndarray, r: int = OUTPUTS_PER_STEP, spec: Optional[np.ndarray] = None,
                   example_id: int = 0) -> Dict[str, np.ndarray]:
    """``DatasetSource._prepare_target.convert`` (datasets/dataset.py:109-144) for one utterance.

    The target is zero-padded at the tail to ``(len // r + 1) * r`` frames only when ``len % r != 0`` (:113-131);
    ``done`` is zeros(len/r - 1) followed by one 1 (:133-135); both loss masks are all ones over the PADDED length
    (:138-139) -- the tail padding inside an utterance is not masked, only the batch padding is.
    """
    mel = np.asarray(mel, np.float32)
    if mel.ndim != 2:
        raise ValueError("mel must be [target_length, num_mels]")
    n = mel.shape[0]
    padded = n if n % r == 0 else (n // r + 1) * r
    out = {"id": np.int64(example_id),
           "mel": np.pad(mel, ((0, padded - n), (0, 0))), "mel_width": np.int64(mel.shape[1]),
           "target_length": np.int64(padded),
           "done": np.concatenate([np.zeros(padded // r - 1, np.float32), np.ones(1, np.float32)]),
           "spec_loss_mask": np.ones(padded, np.float32),
           "binary_loss_mask": np.ones(padded // r, np.float32)}
    if spec is not None:
        spec = np.asarray(spec, np.float32)
        if spec.shape[0] != n:
            raise ValueError("spec and mel must have the same number of frames")
        out["spec"] = np.pad(spec, ((0, padded - n), (0, 0)))
        out["spec_width"] = np.int64(spec.shape[1])
    return out


def prepare_source(ids: Sequence[int], text: str = "", example_id: int = 0) -> Dict[str, object]:
    """``SourceData`` of one utterance (datasets/dataset.py:16-19,102-107): int64 ids and their count."""
    ids = np.asarray(ids, np.int64)
    return {"id": np.int64(example_id), "text": text, "source": ids, "source_length": np.int64(ids.shape[0])}


def keep_by_max_output_length(target_length: int, max_iters: int = MAX_ITERS, r: int = OUTPUTS_PER_STEP) -> bool:
    """``filter_by_max_output_length`` (datasets/dataset.py:172-177).  The filter runs after ``prepare_and_zip``
    (train.py:47), so the length it sees is the one already padded to a multiple of ``r``."""
    return int(target_length) <= max_iters * r


def bucket_key(target_length: int, approx_min_target_length: int = APPROX_MIN_TARGET_LENGTH,
               bucket_width: int = BATCH_BUCKET_WIDTH, num_buckets: int = BATCH_NUM_BUCKETS) -> int:
    """``group_by_batch.key_func`` (datasets/dataset.py:213-216) as written: ``tf.minimum(len - approx_min, 0)`` (not
    maximum), floor-divided by the bucket width, capped at ``num_buckets``.  Every utterance of at least
    ``approx_min_target_length`` frames therefore lands in bucket 0 and shorter ones in negative buckets; the quirk is
    reproduced, not fixed, because it decides which utterances share a batch."""
    v = min(int(target_length) - approx_min_target_length, 0)
    return min(num_buckets, v // bucket_width)          # Python's // floors like tf's integer floordiv


def padded_batch(sources: Sequence[Dict[str, object]], targets: Sequence[Dict[str, np.ndarray]]
                 ) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """``window.padded_batch`` of ``group_by_batch.reduce_func`` (datasets/dataset.py:218-258): pad to the longest of
    the batch with source 0, mel / spec 0, ``done`` 1, both masks 0 (:243-258).  Returns (features, labels) with the
    field names ``model_fn`` reads (models.py:18-21)."""
    if len(sources) != len(targets) or not sources:
        raise ValueError("need the same, non-zero number of sources and targets")
    B = len(sources)
    T_in = max(int(s["source_length"]) for s in sources)
    T_out = max(int(t["target_length"]) for t in targets)
    M = int(targets[0]["mel"].shape[1])
    r_steps = max(int(t["done"].shape[0]) for t in targets)
    feats = {"id": np.array([s["id"] for s in sources], np.int64),
             "text": np.array([s.get("text", "") for s in sources], dtype=object),
             "source": np.zeros((B, T_in), np.int64),
             "source_length": np.array([s["source_length"] for s in sources], np.int64)}
    labels = {"id": np.array([t["id"] for t in targets], np.int64),
              "mel": np.zeros((B, T_out, M), np.float32), "mel_width": np.full(B, M, np.int64),
              "target_length": np.array([t["target_length"] for t in targets], np.int64),
              "done": np.ones((B, r_steps), np.float32),
              "spec_loss_mask": np.zeros((B, T_out), np.float32),
              "binary_loss_mask": np.zeros((B, r_steps), np.float32)}
    has_spec = all("spec" in t for t in targets)
    if has_spec:
        F = int(targets[0]["spec"].shape[1])
        labels["spec"] = np.zeros((B, T_out, F), np.float32)
        labels["spec_width"] = np.full(B, F, np.int64)
    for b, (s, t) in enumerate(zip(sources, targets)):
        n_in, n_out, n_st = int(s["source_length"]), int(t["target_length"]), int(t["done"].shape[0])
        feats["source"][b, :n_in] = s["source"]
        labels["mel"][b, :n_out] = t["mel"]
        labels["done"][b, :n_st] = t["done"]
        labels["spec_loss_mask"][b, :n_out] = t["spec_loss_mask"]
        labels["binary_loss_mask"][b, :n_st] = t["binary_loss_mask"]
        if has_spec:
            labels["spec"][b, :n_out] = t["spec"]
    return feats, labels


def group_by_batch(examples: Iterable[Tuple[Dict[str, object], np.ndarray]], batch_size: int = BATCH_SIZE,
                   r: int = OUTPUTS_PER_STEP, max_iters: int = MAX_ITERS,
                   approx_min_target_length: int = APPROX_MIN_TARGET_LENGTH, bucket_width: int = BATCH_BUCKET_WIDTH,
                   num_buckets: int = BATCH_NUM_BUCKETS, drop_long: bool = True
                   ) -> Iterator[Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]]:
    """``prepare_and_zip().filter_by_max_output_length().group_by_batch()`` (train.py:38-49; datasets/dataset.py:88-100,
    172-177,207-264) over an iterable of (source dict, mel [T, M]) pairs, in order.

    ``tf.contrib.data.group_by_window(key_func, reduce_func, window_size = 5 * batch_size)``: examples are collected per
    key; a key's window is emitted -- as consecutive padded batches of ``batch_size``, the last one possibly smaller --
    when it holds ``5 * batch_size`` examples, and the partial windows are flushed in key order of first appearance at
    the end of the input."""
    window = 5 * batch_size
    pending: Dict[int, List[Tuple[Dict[str, object], Dict[str, np.ndarray]]]] = {}

    def emit(items):
        for i in range(0, len(items), batch_size):
            chunk = items[i:i + batch_size]
            yield padded_batch([c[0] for c in chunk], [c[1] for c in chunk])

    for n, (src, mel) in enumerate(examples):
        tgt = prepare_target(mel, r, example_id=int(src.get("id", n)))
        if drop_long and not keep_by_max_output_length(tgt["target_length"], max_iters, r):
            continue
        key = bucket_key(tgt["target_length"], approx_min_target_length, bucket_width, num_buckets)
        bucket = pending.setdefault(key, [])
        bucket.append((src, tgt))
        if len(bucket) == window:
            yield from emit(bucket)
            pending[key] = []
    for key in list(pending):
        if pending[key]:
            yield from emit(pending[key])
