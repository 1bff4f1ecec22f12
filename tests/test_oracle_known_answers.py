"""CPU tests of the oracle: the known-answer cases derivable from
/root/reference/retriever/hybrid_search.py (the reference has no
tests of its own for this path), the agreement of the two oracle statements (numpy vs C), and the
committed golden fixtures."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _rr(dists, metas, query, top_k=10, symbol_boost=2.0, docs=None):
    ids = [f"id{i}" for i in range(len(dists))]
    return O.rerank_candidates(ids, dists, metas, docs, query, top_k, symbol_boost)


def test_base_score_known_values():                                   # (1) :82
    r = _rr([0.0, 1.0, 4.0], [{}, {}, {}], "q")
    assert [x["score"] for x in r] == [1.0, 0.5, 0.2]


@pytest.mark.parametrize("path,factor", [
    ("pkg/src/a.js", 1.2), ("src/a.js", 1.0), ("a/test/b.js", 0.1), ("a/tests/b.js", 0.1),
    ("a/spec/b.js", 0.1), ("a/__tests__/b.js", 0.1), ("types/x.d.ts", 0.1), ("a/test/src/b.js", 0.1),
    ("A/SRC/b.js", 1.2), ("x/lib/y.ts", 1.2), ("lib/y.ts", 1.0), ("", 1.0),
])
def test_path_class_boost(path, factor):                              # (2) :99-111
    r = _rr([1.0], [{"path": path}], "q")
    assert r[0]["score"] == 0.5 * factor
    assert {0: 1.0, 1: 0.1, 2: 1.2}[O.path_class(path)] == factor


def test_symbol_boost_substring():                                    # (3) :98,114
    q = "Find the implementation of lodash chunk"
    assert _rr([1.0], [{"name": "chunk"}], q, symbol_boost=3.0)[0]["score"] == 1.5
    assert _rr([1.0], [{"name": "Chunk"}], q, symbol_boost=3.0)[0]["score"] == 1.5
    assert _rr([1.0], [{"name": ""}], q, symbol_boost=3.0)[0]["score"] == 0.5
    assert _rr([1.0], [{"name": "a"}], q, symbol_boost=3.0)[0]["score"] == 1.5
    assert _rr([1.0], [{"name": "chunks"}], q, symbol_boost=3.0)[0]["score"] == 0.5
    assert O.symbol_matches("Chunk", q) and not O.symbol_matches("", q)


def test_boosts_compose():                                            # (4)
    r = _rr([1.0], [{"name": "chunk", "path": "p/src/a.js"}], "chunk it", symbol_boost=2.0)
    assert r[0]["score"] == 0.5 * (1.2 * 2.0)


def test_strict_threshold():                                          # (5) :120
    # base 1/(1+99) = 0.01 exactly -> dropped (strict >)
    assert _rr([99.0], [{}], "q") == []
    assert len(_rr([98.0], [{}], "q")) == 1
    # penalised at dist 9: 0.1 * 0.1 is 0.010000000000000002 in fp64, so the reference KEEPS it
    r = _rr([9.0], [{"path": "a/test/b.js"}], "q")
    assert len(r) == 1 and r[0]["score"] == 0.1 * 0.1 and r[0]["score"] > 0.01
    assert _rr([9.5], [{"path": "a/test/b.js"}], "q") == []


def test_stable_sort_and_truncate():                                  # (6) (7) :124-125
    metas = [{}, {}, {}, {}]
    r = _rr([1.0, 1.0, 1.0, 0.5], metas, "q", top_k=3)
    assert [x["id"] for x in r] == ["id3", "id0", "id1"]
    assert len(_rr([1.0] * 20, [{}] * 20, "q", top_k=10)) == 10


def test_metadata_none_and_preview():                                 # (8) (9) :78,:89
    r = O.rerank_candidates(["a"], [1.0], [None], ["line1\nline2" + "x" * 100], "q", 10, 2.0)
    assert r[0]["metadata"] == {}
    assert r[0]["text_preview"] == ("line1 line2" + "x" * 100)[:80]
    r = O.rerank_candidates(["a"], [1.0], [None], None, "q", 10, 2.0)
    assert r[0]["text_preview"] == "" and r[0]["text_full"] is None


def test_missing_and_empty_collection():                              # (10) :54-56,:66-67
    assert O.oracle_search(np.zeros(4, np.float32), "q", None, [], []) == []
    assert O.oracle_search(np.zeros(4, np.float32), "q", np.zeros((0, 4), np.float32), [], []) == []


def test_penalised_neighbour_overtaken_only_inside_2k():              # (11) :62
    # 5 rows on a line; nearest is penalised.  top_k=1 -> only 2 candidates are fetched.
    rows = np.array([[0.1], [0.2], [0.3], [0.4], [0.5]], np.float32)
    metas = [{"path": "a/test/x.js"}, {"path": "a/test/y.js"}, {}, {}, {}]
    ids = [f"r{i}" for i in range(5)]
    q = np.array([0.0], np.float32)
    # top_k=1: candidates r0,r1 (both penalised) -> best penalised wins; r2 is never seen
    assert [x["id"] for x in O.oracle_search(q, "q", rows, ids, metas, top_k=1)] == ["r0"]
    # top_k=2: candidates r0..r3 -> unpenalised r2, r3 overtake
    assert [x["id"] for x in O.oracle_search(q, "q", rows, ids, metas, top_k=2)] == ["r2", "r3"]


def test_collection_name():                                           # api/main.py:77-82
    f = O.collection_name_for
    assert f(None) == "code_chunks" and f("") == "code_chunks" and f(" default ") == "code_chunks"
    assert f("abc-1_2") == "code_chunks__abc-1_2"
    assert f("a b/c") == "code_chunks__a_b_c"
    assert f("x" * 100) == "code_chunks__" + "x" * 64


def test_search_response_filters_after_truncation():                  # api/main.py:191-216
    res = [{"id": "a", "score": 0.9, "metadata": {"name": "f", "kind": "Function", "path": "src/a.js",
                                                  "start_line": 3, "end_line": None}, "text_preview": "p"},
           {"id": "b", "score": 0.2, "metadata": {"path": "lib\\b.js"}, "text_preview": ""}]
    r = O.search_response("q", res)
    assert r["total"] == 2 and r["results"][0]["end_line"] == 0 and r["results"][1]["name"] == ""
    assert O.search_response("q", res, path_prefix="lib/")["total"] == 1
    assert O.search_response("q", res, kind="function")["total"] == 1
    assert O.search_response("q", res, min_score=0.5)["total"] == 1


def test_search_by_symbol_exact():                                    # hybrid_search.py:133-143
    metas = [{"name": "Parser"}, {"name": "parser"}, {"name": "Parser"}, None]
    out = O.search_by_symbol("Parser", ["a", "b", "c", "d"], metas, ["1", "2", "3", "4"], top_k=10)
    assert [x["id"] for x in out] == ["a", "c"] and all(x["score"] == 1.0 for x in out)
    assert O.search_by_symbol("nope", ["a"], [{}], ["1"]) is None


def test_bf16_roundtrip_and_rounding():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.0e-40, 0.0], np.float32)
    b = O.f32_to_bf16_bits(x)
    back = O.bf16_bits_to_f32(b)
    assert back[0] == 1.0 and back[3] == -2.5
    assert back[1] == 1.0            # 1 + 2^-8 is a tie -> even mantissa (1.0)
    assert back[2] == 1.015625       # 1 + 3*2^-8 is a tie -> even mantissa (1 + 2^-6)


@pytest.mark.parametrize("dtype,d,metric", [("f32", 384, "l2"), ("bf16", 768, "l2"), ("f32", 100, "ip"),
                                            ("bf16", 1024, "ip"), ("f32", 7, "l2")])
def test_numpy_and_c_oracle_agree_bitwise(coracle, dtype, d, metric):
    rows = O.synth_unit_rows(2000, d, 7, dtype)
    qs = O.synth_unit_rows(3, d, 8, dtype)
    dc = coracle.distances(rows, qs, metric)
    ids, dist, cnt = coracle.knn(rows, qs, 24, metric)
    for b in range(3):
        dn = O.canonical_distances(qs[b], rows, metric)
        assert np.array_equal(dn.view(np.uint32), dc[b].view(np.uint32))
        i2, d2 = O.knn_candidates(qs[b], rows, 24, metric)
        assert np.array_equal(i2, ids[b]) and np.array_equal(d2.view(np.uint32), dist[b].view(np.uint32))


def test_ties_broken_by_id(coracle):
    rows = np.tile(O.synth_unit_rows(5, 64, 3, "f32"), (40, 1))      # every row appears 40 times
    q = O.synth_unit_rows(1, 64, 4, "f32")
    ids, dist, _ = coracle.knn(rows, q, 20, "l2")
    i2, d2 = O.knn_candidates(q[0], rows, 20, "l2")
    assert np.array_equal(ids[0], i2)
    best = int(ids[0][0]) % 5
    assert list(ids[0]) == [best + 5 * j for j in range(20)]          # ascending ids among exact ties


def test_rerank_numeric_matches_dict_form(coracle):
    rng = np.random.default_rng(0)
    d = np.sort(rng.random(40).astype(np.float32) * 3)
    cls = rng.integers(0, 3, 40).astype(np.uint8)
    mt = (rng.random(40) < 0.3).astype(np.uint8)
    paths = {0: "a/b.js", 1: "a/test/b.js", 2: "a/src/b.js"}
    metas = [{"path": paths[int(c)], "name": "zz" if m else "nomatch"} for c, m in zip(cls, mt)]
    ref = O.rerank_candidates(list(range(40)), [float(x) for x in d], metas, None, "call zz now", 10, 2.5)
    pos, sc = O.rerank_numeric(d, cls, mt, 10, 2.5)
    assert [r["id"] for r in ref] == list(pos) and [r["score"] for r in ref] == list(sc)
    pos2, sc2 = coracle.rerank(d, cls, mt, 10, 2.5)
    assert list(pos2) == list(pos) and list(sc2) == list(sc)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(coracle, path):
    g = np.load(path, allow_pickle=False)
    rows, queries = g["rows"], g["queries"]
    top_k, sb, metric = int(g["top_k"]), float(g["symbol_boost"]), str(g["metric"])
    ids, dist, cnt = coracle.knn(rows, queries, 2 * top_k, metric)
    kc = g["cand_ids"].shape[1]
    assert np.array_equal(ids[:, :kc], g["cand_ids"])
    assert np.array_equal(dist[:, :kc].view(np.uint32), g["cand_d"].view(np.uint32))
    res = coracle.search_numeric(rows, queries, g["classes"], g["match"], top_k, sb, metric)
    for b, (i, s, d) in enumerate(res):
        n = int(g["counts"][b])
        assert len(i) == n
        assert np.array_equal(i, g["ids"][b, :n]) and np.array_equal(s, g["scores"][b, :n])
        assert np.array_equal(d.view(np.uint32), g["dists"][b, :n].view(np.uint32))
