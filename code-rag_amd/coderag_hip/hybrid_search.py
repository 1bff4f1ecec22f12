# -*- coding: utf-8 -*-
"""HybridSearcher: drop-in for /root/reference/retriever/hybrid_search.py:16-146.

Same class name, constructor default, method names, keyword names, defaults, return keys and
"missing collection -> []" behaviour.  What changed underneath: `col.query(...)` on ChromaDB's
CPU HNSW index (hybrid_search.py:60-64) and the Python rerank (:69-125) are one GPU pipeline
behind the C ABI (include/coderag_hip.h); this file only builds the reference's dicts from the
ids/scores the GPU returns.  Added, not altered: `search_batch`, `create_collection`, and an
injected `embed_fn` (Chroma's ONNX MiniLM embedder, hybrid_search.py:23, needs a network fetch
and is out of scope; see embed.py for the offline stand-in).
"""
from __future__ import annotations

import threading
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np

from .collection import GpuCollection, lowered_name_bytes
from ._lib import CragError
from .embed import HashingEmbedder
from . import store


class ChunkCollection:
    """Host-side companion of a GpuCollection: the id / metadata / document tables that
    chromadb keeps in sqlite, plus the name index used by search_by_symbol (:135)."""

    def __init__(self, name: str, dim: int, dtype: str = "f32", metric: str = "l2", device: int = 0):
        self.name = name
        self.gpu = GpuCollection(dim, dtype, metric, device)
        self.ids: List[str] = []
        self.metadatas: List[Optional[dict]] = []
        self.documents: List[Optional[str]] = []
        self._by_name: Dict[Any, List[int]] = {}
        self._bucket: Dict[int, set] = {}
        self._rows_parts: List[np.ndarray] = []  # host copy of the embeddings, for store.save_collection

    def _host_rows(self) -> np.ndarray:
        if not self._rows_parts:
            return np.zeros((0, self.gpu.dim), np.float32)
        return np.concatenate(self._rows_parts, 0)

    def add(self, ids: Sequence[str], embeddings: np.ndarray, metadatas: Optional[Sequence[Optional[dict]]] = None,
            documents: Optional[Sequence[Optional[str]]] = None):
        n = len(ids)
        metadatas = list(metadatas) if metadatas is not None else [None] * n
        documents = list(documents) if documents is not None else [None] * n
        names = [str((m or {}).get("name", "")) for m in metadatas]          # hybrid_search.py:98
        paths = [str((m or {}).get("path", "")) for m in metadatas]          # hybrid_search.py:99
        self.gpu.add(np.asarray(embeddings), names=names, paths=paths)
        self._rows_parts.append(np.array(embeddings, copy=True))
        base = len(self.ids)
        self.ids.extend(ids)
        self.metadatas.extend(metadatas)
        self.documents.extend(documents)
        from .collection import hash_names
        hashes, _ = hash_names(names)
        for i, m in enumerate(metadatas):
            if m is not None and "name" in m:
                self._by_name.setdefault(m["name"], []).append(base + i)
            if hashes[i]:
                self._bucket.setdefault(int(hashes[i]), set()).add(names[i].lower())

    def count(self) -> int:
        return len(self.ids)

    def check_symbol_hashes(self, query: str):
        """Rule out 64-bit hash collisions: every name behind a matched
        hash must really be a substring of the lowered query."""
        q_lower = query.lower()
        for h in self.gpu.symbol_hashes(query):
            for nm in self._bucket.get(int(h), ()):
                if nm not in q_lower:
                    raise CragError(f"symbol hash collision on {nm!r}; refusing to return a wrong boost")


class HybridSearcher:
    def __init__(self, db_path: str = "data/chroma_db", embed_fn: Optional[Callable[[List[str]], np.ndarray]] = None,
                 device: int = 0, dtype: str = "f32"):
        self.db_path = db_path
        self.ef = embed_fn if embed_fn is not None else HashingEmbedder()
        self.device = device
        self.dtype = dtype
        self._cols: Dict[str, ChunkCollection] = {}
        self._lock = threading.Lock()  # the reference mutates _cols from threadpool threads without one

    # ---- collections -----------------------------------------------------------------
    def create_collection(self, name: str, dim: int, dtype: Optional[str] = None, metric: str = "l2",
                          fresh: bool = True) -> ChunkCollection:
        """embed_ingest.py:38-54 (`ensure_collection`): default space l2."""
        with self._lock:
            if not fresh and name in self._cols:
                return self._cols[name]
            col = ChunkCollection(name, dim, dtype or self.dtype, metric, self.device)
            self._cols[name] = col
            return col

    def _get_col(self, name: str) -> Optional[ChunkCollection]:
        with self._lock:
            if name in self._cols:
                return self._cols[name]
        try:
            col = store.load_collection(self.db_path, name, device=self.device)     # :33
        except Exception:
            return None                                                              # :36-39
        with self._lock:
            self._cols[name] = col
        return col

    # ---- search ------------------------------------------------------------------------
    def search(self, query: str, top_k: int = 10, symbol_boost: float = 2.0, include_documents: bool = False,
               collection_name: str = "code_chunks") -> List[Dict[str, Any]]:
        col = self._get_col(collection_name)
        if not col:
            return []                                                                # :54-56
        if col.count() == 0:
            return []                                                                # :66-67
        vec = np.asarray(self.ef([query]), np.float32).reshape(1, -1)
        return self._search_vectors(col, vec, [query], top_k, symbol_boost, include_documents)[0]

    def search_batch(self, queries: Sequence[str], top_k: int = 10, symbol_boost: float = 2.0,
                     include_documents: bool = False, collection_name: str = "code_chunks",
                     embeddings: Optional[np.ndarray] = None) -> List[List[Dict[str, Any]]]:
        col = self._get_col(collection_name)
        if not col or col.count() == 0:
            return [[] for _ in queries]
        vec = np.asarray(embeddings if embeddings is not None else self.ef(list(queries)), np.float32)
        return self._search_vectors(col, vec.reshape(len(queries), -1), list(queries), top_k, symbol_boost,
                                    include_documents)

    def _search_vectors(self, col: ChunkCollection, vec: np.ndarray, queries: List[str], top_k: int,
                        symbol_boost: float, include_documents: bool) -> List[List[Dict[str, Any]]]:
        for q in queries:
            col.check_symbol_hashes(q)
        ids, scores, _dists, counts = col.gpu.search(vec, top_k=top_k, symbol_boost=symbol_boost, query_texts=queries)
        out: List[List[Dict[str, Any]]] = []
        for b in range(len(queries)):
            res = []
            for j in range(int(counts[b])):
                pos = int(ids[b, j])
                doc_text = col.documents[pos] if include_documents else None        # :73
                res.append({
                    "id": col.ids[pos],
                    "score": float(scores[b, j]),
                    "metadata": col.metadatas[pos] or {},                           # :78
                    "text_full": doc_text,
                    "text_preview": (doc_text or "")[:80].replace("\n", " "),       # :89
                })
            out.append(res)
        return out

    def search_by_symbol(self, symbol: str, top_k: int = 10, collection_name: str = "code_chunks"):
        col = self._get_col(collection_name)
        if not col:
            return []                                                                # :130-131
        out = []
        for pos in col._by_name.get(symbol, [])[:top_k]:                             # :135 exact where + limit
            out.append({"id": col.ids[pos], "metadata": col.metadatas[pos], "text_full": col.documents[pos],
                        "score": 1.0})                                               # :142
        if out:
            return out
        return self.search(symbol, top_k=top_k, include_documents=True, collection_name=collection_name)  # :146
