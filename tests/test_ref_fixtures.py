"""Fixtures produced by the REFERENCE'S OWN HybridSearcher (tests/golden/make_ref_fixtures.py ran
/root/reference/retriever/hybrid_search.py in the build container; only its inputs and outputs are
committed, tests/golden/ref/).  They pin hybrid_search.py:66-125 (rerank, filter, stable sort, truncate,
output dict) and :128-146 (search_by_symbol):

  * CPU: oracle/oracle.py reproduces every expected output exactly;
  * GPU: HybridSearcher (HIP path through the C ABI) reproduces every expected output exactly.

The candidate lists behind them are exact squared-L2 k-NN lists; Chroma's own distance numerics and HNSW
recall (hybrid_search.py:60-64) remain parity-unpinned (chromadb is not installed anywhere this runs).
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

REF_DIR = os.path.join(os.path.dirname(__file__), "golden", "ref")


def _load():
    with open(os.path.join(REF_DIR, "ref_rerank.json"), encoding="utf-8") as f:
        fx = json.load(f)
    rows = np.load(os.path.join(REF_DIR, "rows.npz"), allow_pickle=False)
    return fx, {k: rows[k] for k in rows.files}


FX, ROWS = _load()
CASES = FX["cases"]


def _col(key):
    c = FX["collections"].get(key.replace("Araise", "A"))
    if c is None:
        return None
    return c["ids"], c["metadatas"], c["documents"], ROWS[key.replace("Araise", "A")]


def _case_id(c):
    return "%s-%s-%s" % (c["kind"], c["collection"], c.get("top_k", "dflt"))


def test_fixture_file_covers_the_edge_cases():
    kinds = {c["kind"] for c in CASES}
    assert kinds == {"search", "search_defaults", "search_by_symbol"}
    assert any(c["expected"] == [] for c in CASES) and len(CASES) >= 80
    # the reference asked its collection for 2 * top_k results every time (hybrid_search.py:62)
    for c in CASES:
        if c["kind"] == "search" and c["query_calls"]:
            assert [q["n_results"] for q in c["query_calls"]] == [2 * c["top_k"]]
            want_inc = ["metadatas", "documents", "distances"] if c["include_documents"] else ["metadatas", "distances"]
            assert c["query_calls"][0]["include"] == want_inc
    # the strict > 0.01 threshold cases are in there: a distance-99 row is never returned, the penalised
    # distance-9 row (0.1 * 0.1 = 0.010000000000000002) is
    b = [c for c in CASES if c["kind"] == "search" and c["collection"] == "B" and c["top_k"] == 64 and c["symbol_boost"] == 2.5]
    assert b and all("000000000005" not in [r["id"] for r in c["expected"]] for c in b)
    assert any(r["id"] == "000000000006" and r["score"] == 0.1 * 0.1 for c in b for r in c["expected"])


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] in ("search", "search_defaults")], ids=_case_id)
def test_oracle_reproduces_reference_search(case):
    col = _col(case["collection"])
    qv = np.asarray(case["query_vec"], np.float32)
    kw = {} if case["kind"] == "search_defaults" else dict(top_k=case["top_k"], symbol_boost=case["symbol_boost"],
                                                           include_documents=case["include_documents"])
    if col is None:
        got = O.oracle_search(qv, case["query"], None, [], [], None, **kw)
    else:
        ids, metas, docs, rows = col
        got = O.oracle_search(qv, case["query"], rows, ids, metas, docs, **kw)
    assert got == case["expected"]


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "search_by_symbol"], ids=_case_id)
def test_oracle_reproduces_reference_search_by_symbol(case):
    col = _col(case["collection"])
    if col is None:
        assert case["expected"] == []
        return
    ids, metas, docs, rows = col
    got = None if case["collection"] == "Araise" else O.search_by_symbol(case["symbol"], ids, metas, docs, case["top_k"])
    if got is None:  # hybrid_search.py:146: vector query by the symbol text
        got = O.oracle_search(np.asarray(case["query_vec"], np.float32), case["symbol"], rows, ids, metas, docs,
                              top_k=case["top_k"], include_documents=True)
    assert got == case["expected"]


# -------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu_searcher(tmp_path_factory):
    import coderag_hip as ch
    vec = {"v": None}
    hs = ch.HybridSearcher(db_path=str(tmp_path_factory.mktemp("db")), embed_fn=lambda texts: np.tile(vec["v"], (len(texts), 1)))
    for key, c in FX["collections"].items():
        col = hs.create_collection(f"code_chunks__{key}", ROWS[key].shape[1])
        if len(c["ids"]):
            col.add(c["ids"], ROWS[key], c["metadatas"], c["documents"])
        col.gpu.finalize()
    hs._cols["code_chunks"] = hs._cols["code_chunks__A"]
    hs._cols["code_chunks__Araise"] = hs._cols["code_chunks__A"]
    return hs, vec


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] in ("search", "search_defaults")], ids=_case_id)
def test_gpu_searcher_reproduces_reference_search(gpu_searcher, case):
    hs, vec = gpu_searcher
    vec["v"] = np.asarray(case["query_vec"], np.float32)
    if case["kind"] == "search_defaults":
        got = hs.search(case["query"])
    else:
        got = hs.search(case["query"], top_k=case["top_k"], symbol_boost=case["symbol_boost"],
                        include_documents=case["include_documents"], collection_name=f"code_chunks__{case['collection']}")
    assert got == case["expected"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "search_by_symbol" and c["collection"] != "Araise"], ids=_case_id)
def test_gpu_searcher_reproduces_reference_search_by_symbol(gpu_searcher, case):
    hs, vec = gpu_searcher
    vec["v"] = np.asarray(case["query_vec"], np.float32)
    got = hs.search_by_symbol(case["symbol"], top_k=case["top_k"], collection_name=f"code_chunks__{case['collection']}")
    assert got == case["expected"]
