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
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .collection import GpuCollection, lowered_name_bytes, search_multi
from ._lib import CragError
from .config import load_retrieval_config
from .embed import default_embedder
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

    @property
    def device(self) -> int:
        return self.gpu.device

    def _host_rows(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        """The embeddings as the GPU holds them (float32, or uint16 bf16 bit patterns), read back on demand:
        persistence needs them once, so no second copy lives in host memory."""
        return self.gpu.rows_host(first, n)

    def add(self, ids: Sequence[str], embeddings: np.ndarray, metadatas: Optional[Sequence[Optional[dict]]] = None,
            documents: Optional[Sequence[Optional[str]]] = None):
        n = len(ids)
        metadatas = list(metadatas) if metadatas is not None else [None] * n
        documents = list(documents) if documents is not None else [None] * n
        names = [str((m or {}).get("name", "")) for m in metadatas]          # hybrid_search.py:98
        paths = [str((m or {}).get("path", "")) for m in metadatas]          # hybrid_search.py:99
        self.gpu.add(np.asarray(embeddings), names=names, paths=paths)
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

    def check_symbol_hashes(self, query: str) -> np.ndarray:
        """Rule out 64-bit hash collisions: every name behind a matched
        hash must really be a substring of the lowered query.  Returns the query's symbol hashes, which the search
        then takes as they are (the text is hashed once, not once here and once inside the library)."""
        q_lower = query.lower()
        hashes = self.gpu.symbol_hashes(query)
        for h in hashes:
            for nm in self._bucket.get(int(h), ()):
                if nm not in q_lower:
                    raise CragError(f"symbol hash collision on {nm!r}; refusing to return a wrong boost")
        return hashes


def group_requests(keys: Sequence[Optional[Tuple]]) -> List[List[int]]:
    """Requests of one search_many call -> grouped launches.  keys[i] is None for a request that needs no launch
    (missing or empty collection), else (device, dim, dtype, metric, collection identity).  One group per
    (device, dim, dtype, metric) -- a grouped launch needs one geometry on one GPU -- and inside a group the
    requests on one collection are adjacent (they share a pass over its rows), otherwise in request order."""
    groups: Dict[Tuple, List[int]] = {}
    for i, k in enumerate(keys):
        if k is not None:
            groups.setdefault(tuple(k[:4]), []).append(i)
    out = []
    for idx in groups.values():
        first_seen: Dict[Any, int] = {}
        for i in idx:
            first_seen.setdefault(keys[i][4], i)
        out.append(sorted(idx, key=lambda i: (first_seen[keys[i][4]], i)))
    return out


class HybridSearcher:
    def __init__(self, db_path: str = "data/chroma_db", embed_fn: Optional[Callable[[List[str]], np.ndarray]] = None,
                 device: int = 0, dtype: Optional[str] = None, devices: Optional[Sequence[int]] = None,
                 config: Optional[Dict[str, Any]] = None):
        """`devices`: the GPUs this searcher may place collections on.  Whole collections are placed round-robin in
        the order they are created or first loaded (64 workspaces over 8 GPUs = 8 each, so a
        query never needs a collective); `device` is the single-GPU shorthand.  `config`: the retrieval knobs
        (coderag_hip.config.load_retrieval_config(); read from configs/retrieval.yaml when None)."""
        self.db_path = db_path
        self.ef = embed_fn if embed_fn is not None else default_embedder()                # hybrid_search.py:23
        self.devices = [int(d) for d in devices] if devices else [int(device)]
        self.device = self.devices[0]
        self.config = config if config is not None else load_retrieval_config()
        self.dtype = dtype or self.config["dtype"]
        self.metric = self.config["metric"]
        self._cols: Dict[str, ChunkCollection] = {}
        self._placed = 0
        self._lock = threading.Lock()  # the reference mutates _cols from threadpool threads without one
        self._loading: Dict[str, threading.Lock] = {}
        self._pool: Optional[ThreadPoolExecutor] = None

    def _next_device(self) -> int:
        """round-robin placement slot (call with self._lock held)"""
        d = self.devices[self._placed % len(self.devices)]
        self._placed += 1
        return d

    # ---- collections -----------------------------------------------------------------
    def create_collection(self, name: str, dim: int, dtype: Optional[str] = None, metric: Optional[str] = None,
                          fresh: bool = True, device: Optional[int] = None) -> ChunkCollection:
        """embed_ingest.py:38-54 (`ensure_collection`): default space l2."""
        with self._lock:
            if not fresh and name in self._cols:
                return self._cols[name]
            old = self._cols.get(name)
            dev = device if device is not None else (old.device if old is not None else self._next_device())
            col = ChunkCollection(name, dim, dtype or self.dtype, metric or self.metric, dev)
            self._cols[name] = col
            return col

    def _get_col(self, name: str) -> Optional[ChunkCollection]:
        with self._lock:
            if name in self._cols:
                return self._cols[name]
            load_lock = self._loading.setdefault(name, threading.Lock())
        try:
            with load_lock:  # two threads asking for one unloaded collection: the second waits and finds it loaded
                with self._lock:
                    if name in self._cols:
                        return self._cols[name]
                    dev = self.devices[self._placed % len(self.devices)]  # the slot is TAKEN only if the load succeeds
                try:
                    col = store.load_collection(self.db_path, name, device=dev)     # :33
                except Exception:
                    return None                                                      # :36-39
                with self._lock:
                    self._placed += 1
                    self._cols[name] = col
                return col
        finally:
            # names come from the X-Workspace-Id header: a per-name lock must not outlive its load, found or not
            with self._lock:
                self._loading.pop(name, None)

    # ---- search ------------------------------------------------------------------------
    def _knobs(self, top_k, symbol_boost) -> Tuple[int, float]:
        """None -> the configured knob (configs/retrieval.yaml: topK, symbolBoost)"""
        return (int(self.config["topK"]) if top_k is None else int(top_k),
                float(self.config["symbolBoost"]) if symbol_boost is None else float(symbol_boost))

    def search(self, query: str, top_k: int = 10, symbol_boost: float = 2.0, include_documents: bool = False,
               collection_name: str = "code_chunks") -> List[Dict[str, Any]]:
        top_k, symbol_boost = self._knobs(top_k, symbol_boost)
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
        top_k, symbol_boost = self._knobs(top_k, symbol_boost)
        col = self._get_col(collection_name)
        if not col or col.count() == 0:
            return [[] for _ in queries]
        vec = np.asarray(embeddings if embeddings is not None else self.ef(list(queries)), np.float32)
        return self._search_vectors(col, vec.reshape(len(queries), -1), list(queries), top_k, symbol_boost,
                                    include_documents)

    @staticmethod
    def _result_dicts(col: ChunkCollection, ids_b, scores_b, count: int, include_documents: bool) -> List[Dict[str, Any]]:
        res = []
        for j in range(int(count)):
            pos = int(ids_b[j])
            doc_text = col.documents[pos] if include_documents else None            # :73
            res.append({
                "id": col.ids[pos],
                "score": float(scores_b[j]),
                "metadata": col.metadatas[pos] or {},                               # :78
                "text_full": doc_text,
                "text_preview": (doc_text or "")[:80].replace("\n", " "),           # :89
            })
        return res

    def _search_vectors(self, col: ChunkCollection, vec: np.ndarray, queries: List[str], top_k: int,
                        symbol_boost: float, include_documents: bool) -> List[List[Dict[str, Any]]]:
        hashes = [col.check_symbol_hashes(q) for q in queries]
        ids, scores, _dists, counts = search_multi([col.gpu] * len(queries), vec, top_k, symbol_boost, query_hashes=hashes)
        return [self._result_dicts(col, ids[b], scores[b], counts[b], include_documents) for b in range(len(queries))]

    def search_many(self, queries: Sequence[str], collection_names: Sequence[str], top_k: int = 10,
                    symbol_boost: float = 2.0, include_documents: bool = False,
                    embeddings: Optional[np.ndarray] = None) -> List[List[Dict[str, Any]]]:
        """One query per collection, many workspaces at once (api/main.py:77-82: one collection per workspace; a
        threadpool of /search handlers in the reference).  Requests are grouped by the GPU their collection lives
        on; each group is ONE grouped launch (crag_search_multi) and the groups run concurrently, one host thread
        per device -- no cross-GPU traffic, because collections are placed whole.  Missing or empty collections
        give [] like search()."""
        top_k, symbol_boost = self._knobs(top_k, symbol_boost)
        n = len(queries)
        if len(collection_names) != n:
            raise ValueError("one collection name per query")
        cols = [self._get_col(c) for c in collection_names]
        out: List[List[Dict[str, Any]]] = [[] for _ in range(n)]
        live = [i for i in range(n) if cols[i] is not None and cols[i].count() > 0]
        if not live:
            return out
        vec = np.asarray(embeddings if embeddings is not None else self.ef(list(queries)), np.float32).reshape(n, -1)
        hashes = {i: cols[i].check_symbol_hashes(queries[i]) for i in live}
        alive = set(live)
        groups = group_requests([(cols[i].gpu.device, cols[i].gpu.dim, cols[i].gpu.dtype, cols[i].gpu.metric, id(cols[i]))
                                 if i in alive else None for i in range(n)])

        def run(idx: List[int]):
            ids, scores, _d, counts = search_multi([cols[i].gpu for i in idx], vec[idx], top_k, symbol_boost,
                                                   query_hashes=[hashes[i] for i in idx])
            for b, i in enumerate(idx):
                out[i] = self._result_dicts(cols[i], ids[b], scores[b], counts[b], include_documents)

        if len(groups) == 1:
            run(groups[0])
        else:
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=max(2, len(self.devices)), thread_name_prefix="coderag-dev")
            for f in [self._pool.submit(run, g) for g in groups]:
                f.result()
        return out

    def search_by_symbol(self, symbol: str, top_k: int = 10, collection_name: str = "code_chunks"):
        top_k = self._knobs(top_k, None)[0]
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
