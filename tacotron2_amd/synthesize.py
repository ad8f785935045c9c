"""Counterpart of the reference's synthesize.py ``predict`` (:30-69) for the V2 mel path:
ids -> Embedding + EncoderV2 -> free-running decode (stop token) -> PostNetV2 -> mel.
The reference hard-codes batch 1 (:39); the batch dimension is free here (<= 64 per call, one
"reference batch": the stop rule couples its rows).  TFRecord input, the V1 mel->linear PostNet and
Griffin-Lim (:64-69) are outside the path."""
from __future__ import annotations

from typing import Dict, Iterable, List

import numpy as np

from .tacotron_v2 import TacotronV2Model


def text_to_sequence(text: str, eos: int = 1) -> List[int]:
    """datasets/ljspeech.py:154-156 + datasets/corpus.py:80: ord(c) per character, then eos."""
    return [ord(c) for c in text] + [eos]


def synthesize(model: TacotronV2Model, texts: Iterable[str]) -> List[Dict[str, np.ndarray]]:
    texts = list(texts)
    seqs = [text_to_sequence(t) for t in texts]
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    ids = np.zeros((len(seqs), int(lens.max())), dtype=np.int64)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = s
    pred = model.predict({"id": np.arange(len(texts)), "text": texts, "source": ids, "source_length": lens})
    out = []
    for i, t in enumerate(texts):
        out.append({"id": i, "text": t, "mel": pred["mel_postnet"][i].cpu().numpy(),
                    "mel_before_postnet": pred["mel"][i].cpu().numpy(),
                    "alignment": pred["alignment"][i, :lens[i]].cpu().numpy()})
    return out


def synthesize_sharded(model: TacotronV2Model, texts: Iterable[str], world_size: int = None, rank: int = None,
                       batch_size: int = 64) -> Dict[int, Dict[str, np.ndarray]]:
    """Batch-sharded synthesis over the GPUs of a node: the utterances of the whole job are sorted by length
    and cut into "reference batches" of <= batch_size rows (sharding.make_batches), batch b belongs to rank b % world_size
    (sharding.shard_batches), and every rank synthesizes its own batches with its own model replica -- no data-path
    collective; a batch stops when all of ITS rows say stop (helpers.py:90-94), as the reference would given that batch.
    world_size / rank default to the torch.distributed job (1 / 0 without one).  Returns {utterance index: result} for
    this rank's utterances only."""
    from .sharding import make_batches, shard_batches
    if world_size is None or rank is None:
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        world_size = dist.get_world_size() if on else 1
        rank = dist.get_rank() if on else 0
    texts = list(texts)
    seqs = [text_to_sequence(t) for t in texts]
    batches = make_batches([len(s) for s in seqs], batch_size)
    out: Dict[int, Dict[str, np.ndarray]] = {}
    for b in shard_batches(len(batches), world_size, rank):
        idx = batches[b]
        res = synthesize(model, [texts[i] for i in idx])
        for i, r_ in zip(idx, res):
            r_["id"] = int(i)
            out[int(i)] = r_
    return out
