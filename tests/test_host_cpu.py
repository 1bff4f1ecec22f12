"""CPU tests of the host logic and the C-ABI boundary: the library loads and exports every symbol
include/coderag_hip.h declares (no compute call: there is no GPU here), hashing / path
classification / naming agree with the oracle, and the product refuses to run without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import coderag_hip as ch
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "coderag_hip.h")).read()
    declared = set(re.findall(r"\b(crag_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(ch.SYMBOLS), declared ^ set(ch.SYMBOLS)
    lib = ch.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.crag_version() >= 100


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for c in b:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h or 1


def test_name_hashes_are_fnv1a_of_lowered_utf8():
    names = ["chunk", "Chunk", "", "asciiToArray", "größe", "名前", "a"]
    h, l = ch.hash_names(names)
    for n, hv, lv in zip(names, h, l):
        b = n.lower().encode("utf-8")
        assert int(lv) == len(b)
        assert int(hv) == (0 if not b else _fnv1a(b))
    assert h[0] == h[1]


@pytest.mark.parametrize("path", ["pkg/src/a.js", "src/a.js", "a/test/b.js", "a/tests/b.js", "a/spec/b.js",
                                  "a/__tests__/b.js", "types/x.d.ts", "a/test/src/b.js", "A/SRC/b.js", "x/lib/y", "",
                                  None, 12])
def test_classify_path_matches_oracle(path):
    assert ch.classify_path(path) == O.path_class(path)


def test_bf16_cast_matches_oracle():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(5000).astype(np.float32),
                        np.array([0.0, -0.0, 1.00390625, 1.01171875, np.inf, -np.inf, 1e-40], np.float32)])
    assert np.array_equal(ch.f32_to_bf16_bits(x), O.f32_to_bf16_bits(x))


@pytest.mark.parametrize("ws", [None, "", "default", " default ", "abc", "a b/c", "x" * 100, "!!!", "ü-1"])
def test_collection_name_matches_oracle(ws):
    assert ch.get_collection_name(ws) == O.collection_name_for(ws)


class _FakeSearcher:
    def __init__(self, rows):
        self.rows = rows
        self.calls = []

    def search(self, query, top_k=10, symbol_boost=2.0, include_documents=False, collection_name="code_chunks"):
        self.calls.append((query, top_k, symbol_boost, include_documents, collection_name))
        return self.rows


def test_search_route_matches_oracle_response():
    rows = [{"id": "a", "score": 0.9, "metadata": {"name": "f", "kind": "Function", "path": "src/a.js",
                                                   "start_line": 3, "end_line": None}, "text_preview": "p"},
            {"id": "b", "score": 0.2, "metadata": {"path": "lib\\b.js"}, "text_preview": ""}]
    fs = _FakeSearcher(rows)
    for kw in ({}, {"path_prefix": "lib/"}, {"kind": "function"}, {"min_score": 0.5}):
        assert ch.search_route(fs, "q", x_workspace_id="w 1", **kw) == O.search_response("q", rows, **kw)
    assert fs.calls[0] == ("q", 10, 2.0, False, "code_chunks__w_1")


def test_no_gpu_means_loud_failure_not_fallback():
    lib = ch.load()
    if lib.crag_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ch.CragError):
        ch.GpuCollection(16)
    hs = ch.HybridSearcher(db_path="/nonexistent")
    with pytest.raises(ch.CragError):
        hs.create_collection("code_chunks", 16)
    assert hs.search("anything") == []          # missing collection -> [] (hybrid_search.py:54-56)


def test_shard_ranges_partition_rows():
    from coderag_hip.dist import shard_range
    for n in (0, 1, 7, 8, 100, 100_000_001):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_hashing_embedder_is_unit_norm_and_deterministic():
    e = ch.HashingEmbedder()
    a = e(["Find the implementation of lodash chunk", ""])
    b = e(["Find the implementation of lodash chunk"])
    assert a.shape == (2, 384) and a.dtype == np.float32
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-6)
    assert np.array_equal(a[0], b[0])


def test_one_hip_runtime_per_process_whichever_side_loads_first():
    """The library before torch, in a fresh process: both must end up on the same libamdhip64 (a second
    runtime in the process finds no devices)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import coderag_hip as ch\n"
        "ch.load()\n"
        "import torch\n"
        "torch.zeros(1)\n"
        "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "print('\\n'.join(libs))\n" % os.path.join(root, "code-rag_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    libs = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(libs) == 1, libs


def test_chunk_id_and_metadata_sanitising_follow_the_indexer():
    """indexer/chunker.py:158-160 (md5 of path|kind|name|start|end, 12 hex digits) and
    indexer/embed_ingest.py:26-35 (None -> "", non-primitive -> str)."""
    from coderag_hip import store
    assert store.chunk_id({"path": "lodash.js", "kind": "function", "name": "asciiToArray", "start_line": 1, "end_line": 3}) == "4121f2e464e3"
    assert store.chunk_id({"path": "src/größe.ts", "kind": "class", "name": "Größe", "start_line": 10, "end_line": 42}) == "4ff978997a88"
    m = store.sanitize_metadata({"a": None, "b": 1, "c": 1.5, "d": True, "e": "x", "f": [1, 2], "g": {"k": 1}})
    assert m == {"a": "", "b": 1, "c": 1.5, "d": True, "e": "x", "f": "[1, 2]", "g": "{'k': 1}"}


def test_search_routes_over_http_keep_the_reference_shape():
    """api/main.py:143-232 served by coderag_hip.create_app: request model defaults, the X-Workspace-Id header,
    post-filters after truncation, 422 on an empty query, the optional cache keyed by body + collection."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from fastapi.testclient import TestClient
    rows = [{"id": "a", "score": 0.9, "metadata": {"name": "f", "kind": "Function", "path": "src/a.js",
                                                   "start_line": 3, "end_line": None}, "text_preview": "p"},
            {"id": "b", "score": 0.2, "metadata": {"path": "lib\\b.js"}, "text_preview": ""}]

    class Fake(_FakeSearcher):
        def search_by_symbol(self, symbol, top_k=10, collection_name="code_chunks"):
            self.calls.append(("sym", symbol, top_k, collection_name))
            return [{"id": "a", "score": 1.0}]

    class Cache:
        def __init__(self):
            self.d, self.sets = {}, 0

        def get(self, k):
            return self.d.get(k)

        def set(self, k, v, ttl_seconds):
            assert ttl_seconds == 3600
            self.d[k] = v
            self.sets += 1

    fs, cache = Fake(rows), Cache()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        client = TestClient(ch.create_app(fs, cache=cache))
        r = client.post("/search", json={"query": "q"}, headers={"X-Workspace-Id": "w 1"})
        assert r.status_code == 200 and r.json() == O.search_response("q", rows)
        assert fs.calls[-1] == ("q", 10, 2.0, False, "code_chunks__w_1")
        r = client.post("/search", json={"query": "q", "top_k": 5, "symbol_boost": 1.5, "min_score": 0.5})
        assert r.json() == O.search_response("q", rows, min_score=0.5) and fs.calls[-1] == ("q", 5, 1.5, False, "code_chunks")
        n_calls = len(fs.calls)
        assert client.post("/search", json={"query": "q"}, headers={"X-Workspace-Id": "w 1"}).json() == O.search_response("q", rows)
        assert len(fs.calls) == n_calls and cache.sets == 2          # served from the cache, workspace in the key
        assert client.post("/search", json={"query": ""}).status_code == 422
        r = client.get("/search/Parser", params={"top_k": 4}, headers={"X-Workspace-Id": "default"})
        assert r.json() == {"symbol": "Parser", "total": 1, "results": [{"id": "a", "score": 1.0}]}
        assert fs.calls[-1] == ("sym", "Parser", 4, "code_chunks")


# ---- configs/retrieval.yaml (north_star: "the configs/retrieval.yaml knobs (topK, symbolBoost) ... stay unchanged") ----
def test_retrieval_config_file_is_read(tmp_path):
    cfg = ch.load_retrieval_config()                       # this repository's configs/retrieval.yaml
    assert cfg["topK"] == 8 and cfg["symbolBoost"] == 2.0  # clients/vscode/package.json:75-84
    assert cfg["searchDefaults"] == {"top_k": 10, "symbol_boost": 2.0}   # api/main.py:145-146
    p = tmp_path / "retrieval.yaml"
    p.write_text("topK: 5\nsymbolBoost: 3.5\nsearchDefaults:\n  top_k: 7\ndtype: bf16\n")
    cfg = ch.load_retrieval_config(str(p))
    assert (cfg["topK"], cfg["symbolBoost"], cfg["dtype"], cfg["metric"]) == (5, 3.5, "bf16", "l2")
    assert cfg["searchDefaults"] == {"top_k": 7, "symbol_boost": 2.0}
    empty = tmp_path / "empty.yaml"                        # the reference's own file is 0 bytes: defaults
    empty.write_text("")
    assert ch.load_retrieval_config(str(empty))["topK"] == 8
    p.write_text("topK: 500\n")
    with pytest.raises(ValueError):
        ch.load_retrieval_config(str(p))


def test_searcher_takes_its_knobs_from_the_config():
    hs = ch.HybridSearcher(db_path="unused", config={"topK": 5, "symbolBoost": 3.5, "searchDefaults": {"top_k": 7, "symbol_boost": 2.0},
                                                     "metric": "ip", "dtype": "bf16"})
    assert hs._knobs(None, None) == (5, 3.5)               # None -> the configured knob
    assert hs._knobs(10, 2.0) == (10, 2.0)                 # an explicit value wins (the signature defaults are explicit)
    assert (hs.dtype, hs.metric) == ("bf16", "ip")
    import inspect
    sig = inspect.signature(hs.search)                      # the reference's signature stays (hybrid_search.py:42-49)
    assert [(k, v.default) for k, v in sig.parameters.items()] == [
        ("query", inspect.Parameter.empty), ("top_k", 10), ("symbol_boost", 2.0), ("include_documents", False),
        ("collection_name", "code_chunks")]


def test_http_request_model_defaults_and_limits_come_from_the_config():
    from coderag_hip.http_app import make_search_request
    import pydantic
    R = make_search_request({"top_k": 7, "symbol_boost": 2.5})
    r = R(query="x")
    assert (r.top_k, r.symbol_boost) == (7, 2.5)
    for bad in (0, -1, 201):
        with pytest.raises(pydantic.ValidationError):
            R(query="x", top_k=bad)
    with pytest.raises(pydantic.ValidationError):
        R(query="")                                         # api/main.py:144 min_length=1


# ---- whole collections placed over several GPUs (last sentence) ----
def test_collections_are_placed_round_robin_over_the_devices(monkeypatch):
    import coderag_hip.hybrid_search as H

    class FakeCol:
        def __init__(self, name, dim, dtype, metric, device):
            self.name, self.device = name, device

    monkeypatch.setattr(H, "ChunkCollection", FakeCol)
    hs = ch.HybridSearcher(db_path="unused", devices=[0, 1, 2, 3, 4, 5, 6, 7])
    cols = [hs.create_collection(f"code_chunks__ws{i}", 384) for i in range(64)]
    per_dev = {d: sum(1 for c in cols if c.device == d) for d in range(8)}
    assert per_dev == {d: 8 for d in range(8)}              # 64 workspaces over 8 GPUs = 8 each
    assert hs.create_collection("code_chunks__ws3", 384).device == cols[3].device   # fresh re-create stays put
    assert hs.create_collection("code_chunks__pinned", 384, device=5).device == 5
    assert ch.HybridSearcher(db_path="unused", device=2).devices == [2]


def test_default_embedder_follows_the_reference_or_warns(monkeypatch):
    """HybridSearcher() with no embed_fn (the reference's own construction, api/main.py:66-72): Chroma's ONNX MiniLM
    exactly as hybrid_search.py:23 builds it when chromadb imports; otherwise the hashing stand-in WITH a warning."""
    import sys
    import types
    import warnings
    import coderag_hip.embed as E
    # (1) chromadb absent (this container): warning once, hashing stand-in
    monkeypatch.setitem(sys.modules, "chromadb", None)
    monkeypatch.setattr(E, "_warned", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        hs = ch.HybridSearcher(db_path="unused")
        ch.HybridSearcher(db_path="unused")
    assert isinstance(hs.ef, ch.HashingEmbedder)
    msgs = [x for x in w if "NOT a semantic model" in str(x.message)]
    assert len(msgs) == 1 and issubclass(msgs[0].category, RuntimeWarning)
    # (2) chromadb present: its ONNXMiniLM_L6_V2(), called with the list of texts, result as float32 [B, 384]
    calls = []

    class FakeOnnx:
        def __call__(self, texts):
            calls.append(list(texts))
            return [[float(len(t))] * 384 for t in texts]

    chroma = types.ModuleType("chromadb")
    utils = types.ModuleType("chromadb.utils")
    ef = types.ModuleType("chromadb.utils.embedding_functions")
    ef.ONNXMiniLM_L6_V2 = FakeOnnx
    utils.embedding_functions = ef
    chroma.utils = utils
    for name, mod in (("chromadb", chroma), ("chromadb.utils", utils), ("chromadb.utils.embedding_functions", ef)):
        monkeypatch.setitem(sys.modules, name, mod)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # no warning on this branch
        hs = ch.HybridSearcher(db_path="unused")
    v = hs.ef(["ab", "abcd"])
    assert calls == [["ab", "abcd"]] and v.dtype == np.float32 and v.shape == (2, 384) and v[1, 0] == 4.0
    # an explicit embed_fn is taken as it is
    f = lambda texts: np.zeros((len(texts), 384), np.float32)
    assert ch.HybridSearcher(db_path="unused", embed_fn=f).ef is f


def test_failed_collection_loads_leave_no_state_behind(monkeypatch, tmp_path):
    """Collection names come from the X-Workspace-Id header: a name that is not on disk must leave neither a per-name
    load lock nor a placement slot behind (hybrid_search.py:36-39 swallows the error and answers [])."""
    hs = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=ch.HashingEmbedder(), devices=[0, 1])
    for i in range(50):
        assert hs.search("x", collection_name=f"code_chunks__nobody{i}") == []
    assert hs._loading == {} and hs._placed == 0 and hs._cols == {}


def test_group_requests_one_launch_per_device_and_geometry():
    k = lambda dev, name, dim=1024, dt="bf16": (dev, dim, dt, "l2", name)
    keys = [k(0, "a"), k(1, "b"), None, k(0, "c"), k(0, "a"), k(1, "d"), k(1, "b"), k(0, "e", 384, "f32")]
    groups = ch.group_requests(keys)
    assert sorted(map(tuple, groups)) == [(0, 4, 3), (1, 6, 5), (7,)]   # same collection adjacent, one group per geometry
    assert ch.group_requests([None, None]) == []
    flat = sorted(i for g in groups for i in g)
    assert flat == [0, 1, 3, 4, 5, 6, 7]                    # every live request exactly once


# ---- the reference's mini eval metrics (eval/run_eval.py), /search half ----
def test_eval_metrics_reproduce_the_committed_report():
    """Hit@K / MRR@K / percentile as eval/run_eval.py computes them.  A ranking in which the three queries of
    eval/qa_min.jsonl find their gold keys at ranks 1, 2 and 1 gives the reference's committed numbers
    (eval/results/report.json: hit@5 1.0, mrr@5 0.8333...)."""
    from coderag_hip import eval as E
    qa = [{"id": "q1", "query": "What does asciiToArray do?", "expect": {"hit": ["lodash.js:asciiToArray"]}},
          {"id": "q2", "query": "Find the implementation of lodash chunk", "expect": {"hit": ["lodash.js:chunk", "underscore.js:chunk"]}},
          {"id": "q3", "query": "How to count unicode symbols length in lodash?", "expect": {"hit": ["lodash.js:stringSize", "lodash.js:hasUnicode"]}}]
    answers = {"q1": [("lodash.js", "asciiToArray"), ("lodash.js", "x")],
               "q2": [("lodash.js", "baseSlice"), ("underscore.js", "chunk"), ("lodash.js", "chunk")],
               "q3": [("lodash.js", "hasUnicode"), ("lodash.js", "stringSize")]}
    by_query = {x["query"]: x["id"] for x in qa}

    def search_fn(query, top_k, sb):
        assert (top_k, sb) == (5, 2.0)
        return {"results": [{"path": p, "name": n} for p, n in answers[by_query[query]]]}

    rep = E.evaluate_search(search_fn, qa, top_k=5)
    assert rep["n"] == 3 and rep["search"]["hit@k"] == 1.0 and rep["search"]["mrr@k"] == (1 + 0.5 + 1) / 3
    assert abs(rep["search"]["mrr@k"] - 0.8333333333333334) < 1e-15      # eval/results/report.json:6
    assert E.mrr_at_k(["a:b"], ["x:y"]) == 0.0 and E.hit_at_k([], ["x:y"]) == 0.0
    assert E.percentile([10.0, 20.0, 30.0, 40.0], 0.5) == 25.0 and E.percentile([], 0.95) == 0.0 and E.percentile([7.0], 0.95) == 7.0
    boom = E.evaluate_search(lambda q, k, s: (_ for _ in ()).throw(RuntimeError("down")), qa[:1])
    assert boom["search"]["hit@k"] == 0.0                                  # an error counts as no results


def test_isa_guards_catch_the_gemv_faults_of_round_2(tmp_path):
    """build.py refuses a GEMV whose ISA shows one of the two faults found in round 2: a swap
    too close behind v_dot2c, FLAT row loads, waits that drain every load.  Checked on doctored copies of the listing
    the build emitted for the library in the tree."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("crag_build", os.path.join(ROOT, "code-rag_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    src = os.path.join(ROOT, "code-rag_amd", "build", "crag_scan_gemv.s")
    if not os.path.exists(src):
        pytest.skip("no ISA listing: the library was not built in this tree")
    text = open(src).read()
    b.check_gemv_isa(src)
    doctored = {
        "nop": text.replace("s_nop 4", "s_nop 1"),
        "flat": re.sub(r"global_load_dwordx4 (v\[\d+:\d+\]), (v\[\d+:\d+\]), off( nt)?", r"flat_load_dwordx4 \1, \2", text, count=1),
        "drain": re.sub(r"s_waitcnt vmcnt\(\d+\)", "s_waitcnt vmcnt(0)", text),
    }
    for name, t in doctored.items():
        assert t != text, name
        f = tmp_path / f"{name}.s"
        f.write_text(t)
        with pytest.raises(RuntimeError):
            b.check_gemv_isa(str(f))


@pytest.mark.parametrize("gen,inc", [("gen_q16_kloop.py", "crag_q16_kloop.inc"), ("gen_qsplit_kloop.py", "crag_qsplit_kloop.inc")])
def test_generated_tile_statements_are_what_their_generators_emit(gen, inc):
    """The inline-asm tile statements the scan kernels include are generated files: the committed text must be the
    generator's output with its default settings (an ablation build left behind would ship a different kernel)."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "code-rag_amd", "csrc")
    env = {k: v for k, v in os.environ.items() if not k.startswith("Q16_")}
    out = subprocess.run([sys.executable, os.path.join(csrc, gen)], capture_output=True, text=True, check=True, env=env).stdout
    assert out == open(os.path.join(csrc, inc)).read()
