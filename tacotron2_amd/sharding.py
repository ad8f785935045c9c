"""Batch sharding of utterances over the GPUs of a node.

Utterances are independent, so inference needs no data-path collective: every rank takes its own
"reference batches" (<= 64 rows each; the stop rule of helpers.py:90-94 couples the rows of a batch,
so a shard stops exactly as the reference would given that shard as its batch).  Sorting by length
before dealing limits padding, the reference's analogue being its bucketing by target length
(datasets/dataset.py:207-264)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def make_batches(lengths: Sequence[int], batch_size: int = 64) -> List[np.ndarray]:
    """Indices of the utterances of each batch: sorted by length (stable), cut into batch_size rows."""
    order = np.argsort(np.asarray(lengths), kind="stable")
    return [order[i:i + batch_size] for i in range(0, len(order), batch_size)]


def shard_batches(n_batches: int, world_size: int, rank: int) -> List[int]:
    """Batch b goes to rank b % world_size (round robin keeps the per-rank padding similar)."""
    return list(range(rank, n_batches, world_size))


def pad_ids(seqs: Sequence[Sequence[int]]):
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    ids = np.zeros((len(seqs), int(lens.max())), dtype=np.int64)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = s
    return ids, lens
