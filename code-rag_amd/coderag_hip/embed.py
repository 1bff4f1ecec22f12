# -*- coding: utf-8 -*-
"""Offline stand-in for the query/document embedder.

The reference embeds with Chroma's `embedding_functions.ONNXMiniLM_L6_V2()`
(/root/reference/retriever/hybrid_search.py:23, indexer/embed_ingest.py:48), which downloads
the all-MiniLM-L6-v2 ONNX archive on first use.  That model is out of scope for this path
(north_star starts at vectors) and cannot be fetched here, so HybridSearcher takes an injected
`embed_fn: List[str] -> float32[B, D]`.  This class is the default: a deterministic character
n-gram feature hasher, 384-d and L2-normalised like MiniLM's output.  It is NOT a semantic
model; plug the real embedder in for retrieval quality.

`default_embedder()` is what `HybridSearcher()` takes when no `embed_fn` is given -- the reference constructs its
searcher with no arguments (api/main.py:66-72): Chroma's own ONNX MiniLM when chromadb imports, built exactly as
hybrid_search.py:23 builds it, so that a one-line swap keeps the reference's embeddings; otherwise the hashing
stand-in, with a warning (once per process) that the results are not semantic.
"""
from __future__ import annotations

import warnings
import zlib
from typing import Callable, List

import numpy as np

_warned = False


def default_embedder() -> Callable[[List[str]], np.ndarray]:
    global _warned
    try:
        from chromadb.utils import embedding_functions  # hybrid_search.py:12
        onnx = embedding_functions.ONNXMiniLM_L6_V2()   # hybrid_search.py:23
    except Exception:  # not installed, or its model cannot be fetched: the offline stand-in, loudly
        if not _warned:
            _warned = True
            warnings.warn("coderag_hip.HybridSearcher: no embed_fn given and chromadb's ONNXMiniLM_L6_V2 is not available; "
                          "falling back to HashingEmbedder, a character n-gram feature hasher that is NOT a semantic model. "
                          "Pass embed_fn=<your embedder> for meaningful retrieval.", RuntimeWarning, stacklevel=3)
        return HashingEmbedder()
    return lambda texts: np.asarray(onnx(list(texts)), np.float32)


class HashingEmbedder:
    def __init__(self, dim: int = 384, ngram: int = 3):
        self.dim = dim
        self.ngram = ngram

    def __call__(self, texts: List[str]) -> np.ndarray:
        out = np.zeros((len(texts), self.dim), np.float32)
        for i, t in enumerate(texts):
            s = " " + str(t).lower() + " "
            b = s.encode("utf-8", "ignore")
            for n in range(1, self.ngram + 1):
                for j in range(0, max(len(b) - n + 1, 0)):
                    h = zlib.crc32(b[j:j + n])
                    out[i, h % self.dim] += 1.0 if (h >> 31) & 1 else -1.0
            nrm = float(np.linalg.norm(out[i]))
            if nrm > 0:
                out[i] /= nrm
            else:
                out[i, 0] = 1.0
        return out
