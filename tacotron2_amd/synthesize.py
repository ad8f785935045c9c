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
