# -*- coding: utf-8 -*-
"""On-disk collection format: what stands where Chroma's
`data/chroma_db` persistence stands in the reference (indexer/embed_ingest.py:77-105).

    <db_path>/<collection_name>/meta.json     {"dim", "dtype", "metric", "count"}
    <db_path>/<collection_name>/rows.npy      [count, dim] float32, or uint16 bf16 bit patterns
    <db_path>/<collection_name>/chunks.jsonl  one {"id", "meta", "text"} per row (the chunker's
                                              record shape, indexer/chunker.py:270-278)

`ingest_chunks` is the run_ingest of embed_ingest.py:57-132 with the embeddings produced by the
injected embed_fn; metadata is sanitised like embed_ingest.py:26-35.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Any, Callable, Dict, Iterable, List, Optional

import numpy as np


ROWS_PER_SLICE = 262144  # rows moved between disk, host and device at a time


def sanitize_metadata(meta: Dict[str, Any]) -> Dict[str, Any]:
    """Taken from the reference (indexer/embed_ingest.py:26-35) as it stands: its output is the metadata contract."""
    out = {}
    for k, v in meta.items():
        if v is None:
            out[k] = ""
        elif isinstance(v, (bool, int, float, str)):
            out[k] = v
        else:
            out[k] = str(v)
    return out


def chunk_id(meta: Dict[str, Any]) -> str:
    """The chunker's id of a chunk record (indexer/chunker.py:158-160, 277): the first 12 hex digits of
    md5("<path>|<kind>|<name>|<start_line>|<end_line>").  Used for records that arrive without an "id"."""
    key = "|".join(str(meta.get(k, "")) for k in ("path", "kind", "name", "start_line", "end_line"))
    return hashlib.md5(key.encode("utf-8")).hexdigest()[:12]


def iter_jsonl(path: str):
    """Taken from the reference (indexer/embed_ingest.py:18-24) as it stands."""
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def save_collection(db_path: str, col) -> str:
    d = os.path.join(db_path, col.name)
    os.makedirs(d, exist_ok=True)
    # rows come back from the device in slices and go straight into the .npy file: no whole-matrix host copy
    n, dim = col.count(), col.gpu.dim
    out = np.lib.format.open_memmap(os.path.join(d, "rows.npy"), mode="w+",
                                    dtype=np.float32 if col.gpu.dtype == "f32" else np.uint16, shape=(n, dim))
    for lo in range(0, n, ROWS_PER_SLICE):
        m = min(ROWS_PER_SLICE, n - lo)
        out[lo:lo + m] = col._host_rows(lo, m)
    out.flush()
    del out
    with open(os.path.join(d, "chunks.jsonl"), "w", encoding="utf-8") as f:
        for cid, m, doc in zip(col.ids, col.metadatas, col.documents):
            f.write(json.dumps({"id": cid, "meta": m, "text": doc}, ensure_ascii=False) + "\n")
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump({"dim": col.gpu.dim, "dtype": col.gpu.dtype, "metric": col.gpu.metric, "count": col.count()}, f)
    return d


def load_collection(db_path: str, name: str, device: int = 0):
    from .hybrid_search import ChunkCollection
    d = os.path.join(db_path, name)
    with open(os.path.join(d, "meta.json")) as f:
        meta = json.load(f)
    rows = np.load(os.path.join(d, "rows.npy"), mmap_mode="r", allow_pickle=False)  # paged in slice by slice
    col = ChunkCollection(name, meta["dim"], meta["dtype"], meta.get("metric", "l2"), device)
    ids, metas, docs = [], [], []
    done = 0

    def flush():
        nonlocal done
        if ids:
            col.add(list(ids), np.ascontiguousarray(rows[done:done + len(ids)]), list(metas), list(docs))
            done += len(ids)
            ids.clear()
            metas.clear()
            docs.clear()

    for obj in iter_jsonl(os.path.join(d, "chunks.jsonl")):
        ids.append(obj["id"])
        metas.append(obj.get("meta"))
        docs.append(obj.get("text"))
        if len(ids) >= ROWS_PER_SLICE:
            flush()
    flush()
    if done != rows.shape[0]:
        raise ValueError("rows.npy and chunks.jsonl disagree")
    col.gpu.finalize()
    return col


def ingest_chunks(searcher, chunks_path: str, collection: str = "code_chunks", batch_size: int = 100,
                  fresh: bool = False, persist: bool = True) -> Dict[str, Any]:
    """embed_ingest.py:57-132: JSONL of {id, text, meta} -> collection, batches of `batch_size`."""
    assert os.path.exists(chunks_path), f"chunks file not found: {chunks_path}"
    probe = np.asarray(searcher.ef(["probe"]), np.float32)
    col = searcher.create_collection(collection, probe.shape[-1], fresh=fresh)
    ids: List[str] = []
    docs: List[str] = []
    metas: List[Dict[str, Any]] = []
    total = 0

    def flush():
        if not ids:
            return
        emb = np.asarray(searcher.ef(list(docs)), np.float32)
        col.add(list(ids), emb, list(metas), list(docs))
        ids.clear()
        docs.clear()
        metas.clear()

    for obj in iter_jsonl(chunks_path):
        ids.append(obj.get("id") or chunk_id(obj.get("meta", {})))
        docs.append(obj["text"])
        metas.append(sanitize_metadata(obj.get("meta", {})))
        total += 1
        if len(ids) >= int(batch_size):
            flush()
    flush()
    col.gpu.finalize()
    if persist:
        save_collection(searcher.db_path, col)
    return {"chunks": chunks_path, "db": searcher.db_path, "collection": collection, "fresh": fresh,
            "batch_size": int(batch_size), "input_count": total, "collection_count": col.count()}
