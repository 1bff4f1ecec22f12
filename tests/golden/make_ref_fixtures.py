#!/usr/bin/env python3
"""Generates tests/golden/ref/ref_rerank.json + tests/golden/ref/rows.npz by RUNNING THE REFERENCE'S OWN
HybridSearcher (imported from its mounted path, /root/reference/retriever/hybrid_search.py -- nothing of
it is copied into this repository or shipped to the GPU box).  BUILD CONTAINER ONLY.

What this pins, and what it cannot:

  * hybrid_search.py:66-125 (empty-result guard, base score, path/symbol boost, > 0.01 filter, stable sort,
    truncate, the output dict incl. text_preview) and :128-146 (search_by_symbol) are executed verbatim by
    the reference's own interpreter-level code on the inputs below; the expected outputs in the fixture
    are ITS outputs.  tests/test_ref_fixtures.py checks oracle/oracle.py against them on the CPU and
    HybridSearcher (HIP path, through the C ABI) against them on the GPU.
  * hybrid_search.py:60-64 (`col.query`: MiniLM embedding, Chroma's float32 distance arithmetic, HNSW
    recall) lives in chromadb==1.3.0, which is not installed here.  The module below stands where the
    `chromadb` package stands ONLY so that the reference file's three import lines resolve and so that
    `col.query` / `col.get` can hand the reference code a candidate list; the candidate lists themselves
    are exact squared-L2 k-NN lists (oracle arithmetic).  That boundary stays PARITY UNPINNED.

    python tests/golden/make_ref_fixtures.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference/retriever/hybrid_search.py"


# ---------------------------------------------------------------------------------------------------
# an in-memory stand-in for the chromadb objects the reference file touches (import lines :10-12,
# PersistentClient :18, get_collection :33, Collection.query :60, Collection.get :135)
class FakeCollection:
    def __init__(self, ids, rows, metas, docs, get_raises=False):
        self.ids, self.rows, self.metas, self.docs = ids, rows, metas, docs
        self.get_raises = get_raises
        self.query_vec = None       # set per case: what the embedder would have produced
        self.calls = []

    def query(self, query_texts, n_results, include):
        self.calls.append({"n_results": n_results, "include": list(include)})
        assert len(query_texts) == 1
        if len(self.ids) == 0:
            return {"ids": [[]], "distances": [[]], "metadatas": [[]], "documents": [[]]}
        pos, d32 = O.knn_candidates(self.query_vec, self.rows, n_results, "l2")
        out = {"ids": [[self.ids[int(p)] for p in pos]],
               "distances": [[float(x) for x in d32]],            # Chroma returns float32-origin Python floats
               "metadatas": [[self.metas[int(p)] for p in pos]]}
        if "documents" in include:
            out["documents"] = [[self.docs[int(p)] for p in pos]]
        return out

    def get(self, where, limit, include):
        if self.get_raises:
            raise RuntimeError("where clause rejected")
        (key, val), = where.items()
        hit = [i for i, m in enumerate(self.metas) if (m or {}).get(key) == val][:limit]
        return {"ids": [self.ids[i] for i in hit], "metadatas": [self.metas[i] for i in hit],
                "documents": [self.docs[i] for i in hit]}


class FakeClient:
    collections = {}

    def __init__(self, path=None, settings=None):
        self.path = path

    def get_collection(self, name, embedding_function=None):
        if name not in FakeClient.collections:
            raise ValueError(f"Collection {name} does not exist.")
        return FakeClient.collections[name]


def import_reference():
    chromadb = types.ModuleType("chromadb")
    chromadb.PersistentClient = FakeClient
    chromadb.Collection = FakeCollection
    config = types.ModuleType("chromadb.config")
    config.Settings = lambda **kw: kw
    utils = types.ModuleType("chromadb.utils")
    ef = types.ModuleType("chromadb.utils.embedding_functions")
    ef.ONNXMiniLM_L6_V2 = lambda: None
    utils.embedding_functions = ef
    chromadb.config, chromadb.utils = config, utils
    sys.modules.update({"chromadb": chromadb, "chromadb.config": config, "chromadb.utils": utils,
                        "chromadb.utils.embedding_functions": ef})
    spec = importlib.util.spec_from_file_location("ref_hybrid_search", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------------------------------------------
NAMES = ["chunk", "Chunk", "asciiToArray", "stringSize", "hasUnicode", "Parser", "baseFlatten", "debounce", "map", "a",
         "größe", "İstanbul", "isArrayLike", "castPath", "", None, 42, "ǅungla", "ﬂatten", "x" * 70]
DIRS = ["lodash/src/", "src/", "pkg/test/", "lib/", "a/lib/", "types/", "x/__tests__/", "docs/spec/", "", "A/SRC/",
        "w/tests/", "a/test/src/", "C:\\repo\\src\\"]


def make_chunks(n, rng):
    ids, metas, docs = [], [], []
    for i in range(n):
        nm = NAMES[int(rng.integers(0, len(NAMES)))]
        p = DIRS[int(rng.integers(0, len(DIRS)))] + (f"f{i % 17}.d.ts" if i % 11 == 0 else f"f{i % 17}.js")
        m = {"name": nm, "path": p, "kind": "Function" if i % 2 else "Class", "start_line": i, "end_line": i + 3,
             "ast_path": f"program/fn{i}", "exported": bool(i % 3)}
        if i % 13 == 0:
            del m["name"]                               # missing key -> "" (:98)
        if i % 19 == 0:
            del m["path"]
        if i % 23 == 0:
            m["path"] = None                            # str(None).lower() == "none"
        metas.append(None if i % 29 == 0 else m)        # metas[i] or {} (:78)
        ids.append(f"{i:012x}")
        docs.append(None if i % 31 == 0 else f"function {nm}() {{\n  return {i};\n}}" + " // pad" * (i % 20))
    return ids, metas, docs


QUERIES = ["Find the implementation of lodash chunk", "What does asciiToArray do?",
           "How to count unicode symbols length in lodash? stringSize hasUnicode", "GRÖSSE größe Parser i̇stanbul",
           "zzz", "none 42 " + "x" * 70, "a", "", "ǆungla ﬂatten MAP", "İSTANBUL"]


def main():
    ref = import_reference()
    rng = np.random.default_rng(20261111)
    d = 32
    cols = {}
    # A: unit rows with every vector stored three times (exact distance ties: the stable sort and the
    #    (distance, id) candidate order decide)
    base = O.synth_unit_rows(140, d, 11, "f32")
    rowsA = np.concatenate([base, base, base], 0)
    # B: rows of integer coordinates around a zero query: distances are exact integers (9 -> 0.1, 99 -> 0.01 ...)
    rowsB = np.zeros((64, d), np.float32)
    for i in range(64):
        rowsB[i, i % d] = float(i % 11)
        rowsB[i, (i + 1) % d] = 3.0 if i % 2 else 0.0
        rowsB[i, (i + 5) % d] = 9.0 if i % 7 == 0 else 0.0
    rowsB[5:7] = 0.0
    rowsB[5, :3] = (9.0, 3.0, 3.0)                      # |x|^2 = 99 -> base exactly 0.01: dropped (strict >)
    rowsB[6, :3] = (3.0, 0.0, 0.0)                      # |x|^2 = 9
    # C: fewer rows than 2 * top_k
    rowsC = O.synth_unit_rows(7, d, 13, "f32")
    for key, rows in (("A", rowsA), ("B", rowsB), ("C", rowsC), ("E", np.zeros((0, d), np.float32))):
        ids, metas, docs = make_chunks(len(rows), rng)
        if key == "B":                                  # pin the threshold cases' metadata
            metas[5] = {"name": "plain", "path": "docs/readme.md"}
            metas[6] = {"name": "pen", "path": "a/test/b.js"}     # 0.1 * 0.1 = 0.010000000000000002 > 0.01: kept
        cols[key] = {"ids": ids, "metas": metas, "docs": docs, "rows": rows}
    FakeClient.collections = {f"code_chunks__{k}": FakeCollection(c["ids"], c["rows"], c["metas"], c["docs"])
                              for k, c in cols.items()}
    FakeClient.collections["code_chunks__Araise"] = FakeCollection(cols["A"]["ids"], cols["A"]["rows"], cols["A"]["metas"],
                                                                   cols["A"]["docs"], get_raises=True)
    hs = ref.HybridSearcher(db_path="unused")
    qvecs = O.synth_unit_rows(len(QUERIES), d, 17, "f32")
    cases = []

    def run_search(ckey, qi, top_k, sb, inc, qvec=None):
        col = FakeClient.collections.get(f"code_chunks__{ckey}")
        v = qvecs[qi] if qvec is None else qvec
        if col is not None:
            col.query_vec = v
            col.calls.clear()
        got = hs.search(QUERIES[qi], top_k=top_k, symbol_boost=sb, include_documents=inc,
                        collection_name=f"code_chunks__{ckey}")
        cases.append({"kind": "search", "collection": ckey, "query": QUERIES[qi], "query_vec": [float(x) for x in v],
                      "top_k": top_k, "symbol_boost": sb, "include_documents": inc,
                      "query_calls": list(col.calls) if col is not None else None, "expected": got})

    for qi in range(len(QUERIES)):
        for top_k, sb, inc in ((1, 2.0, False), (8, 2.0, False), (10, 2.5, True), (100, 2.0, False), (8, 0.3, True),
                               (10, 0.004, False)):
            run_search("A", qi, top_k, sb, inc)
    zero = np.zeros(d, np.float32)
    for qi in (0, 5, 7):
        for top_k, sb, inc in ((8, 2.0, True), (10, 0.3, False), (64, 2.5, False), (10, 0.004, True)):
            run_search("B", qi, top_k, sb, inc, zero)
    for qi in (0, 4):
        run_search("C", qi, 10, 2.0, True)
        run_search("C", qi, 3, 2.0, False)
    run_search("E", 0, 10, 2.0, False)                  # empty collection -> [] (:66-67)
    run_search("nope", 0, 10, 2.0, False)               # missing collection -> [] (:54-56)
    # defaults of the signature (:42-49)
    FakeClient.collections["code_chunks"] = FakeClient.collections["code_chunks__A"]
    FakeClient.collections["code_chunks"].query_vec = qvecs[0]
    FakeClient.collections["code_chunks"].calls.clear()
    got = hs.search(QUERIES[0])
    cases.append({"kind": "search_defaults", "collection": "A", "query": QUERIES[0], "query_vec": [float(x) for x in qvecs[0]],
                  "query_calls": list(FakeClient.collections["code_chunks"].calls), "expected": got})
    # search_by_symbol (:128-146): exact branch, limit, fallback on no match, fallback when get raises, missing collection
    for ckey, sym, top_k, qi in (("A", "Parser", 4, None), ("A", "chunk", 10, None), ("A", "parser", 5, 3),
                                 ("Araise", "Parser", 5, 3), ("nope", "Parser", 5, None), ("A", 42, 3, None)):
        col = FakeClient.collections.get(f"code_chunks__{ckey}")
        v = qvecs[qi] if qi is not None else qvecs[0]
        if col is not None:
            col.query_vec = v
        got = hs.search_by_symbol(sym, top_k=top_k, collection_name=f"code_chunks__{ckey}")
        cases.append({"kind": "search_by_symbol", "collection": ckey, "symbol": sym, "top_k": top_k,
                      "query_vec": [float(x) for x in v], "expected": got})

    out = {"generated_by": "tests/golden/make_ref_fixtures.py running /root/reference/retriever/hybrid_search.py",
           "collections": {k: {"ids": c["ids"], "metadatas": c["metas"], "documents": c["docs"]} for k, c in cols.items()},
           "cases": cases}
    os.makedirs(os.path.join(HERE, "ref"), exist_ok=True)
    with open(os.path.join(HERE, "ref", "ref_rerank.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    np.savez_compressed(os.path.join(HERE, "ref", "rows.npz"), **{k: c["rows"] for k, c in cols.items()})
    print(f"{len(cases)} cases written; non-empty results: {sum(1 for c in cases if c['expected'])}")


if __name__ == "__main__":
    main()
