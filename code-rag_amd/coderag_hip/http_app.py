# -*- coding: utf-8 -*-
"""The retrieval routes of api/main.py as a FastAPI application.

POST /search (api/main.py:143-220): body {query, top_k=10, symbol_boost=2.0, path_prefix, kind, min_score},
header X-Workspace-Id; an empty query is a 422.  GET /search/{symbol} (api/main.py:222-232).  Both are sync
handlers, so they run on the framework's threadpool like the reference's; the searcher is re-entrant.

The cache stands where api/cache.py's Redis helpers stand and is optional: any object with
get(key) -> response dict or None and set(key, response_dict, ttl_seconds).  The key is the request body plus
the resolved collection name, as in api/main.py:175-176.

(No `from __future__ import annotations` here: FastAPI resolves the handlers' annotations at decoration.)
"""
import json
from typing import List, Optional

from fastapi import FastAPI, Header, Query
from pydantic import BaseModel, Field

from .api import get_collection_name, search_route


def make_search_request(defaults: dict):
    """The reference's request model (api/main.py:143-149) with its two defaults taken from configs/retrieval.yaml
    `searchDefaults` (10 and 2.0 there as well).  top_k is validated here: the kernels serve 1..200
    (CRAG_MAX_TOP_K); the reference has no upper limit, so a larger value is a 422 here instead of a 500."""

    class SearchRequest(BaseModel):
        query: str = Field(..., min_length=1)
        top_k: int = Field(int(defaults.get("top_k", 10)), ge=1, le=200)
        symbol_boost: float = float(defaults.get("symbol_boost", 2.0))
        path_prefix: Optional[str] = None
        kind: Optional[str] = None
        min_score: Optional[float] = None

    return SearchRequest


class SearchResult(BaseModel):
    id: str
    score: float
    name: str = ""
    kind: str = ""
    path: str = ""
    start_line: int = 0
    end_line: int = 0
    text_preview: str = ""


class SearchResponse(BaseModel):
    query: str
    total: int
    results: List[SearchResult]


def build_app(searcher, cache=None) -> FastAPI:
    app = FastAPI(title="code-rag retrieval (MI355X scoring path)")
    cfg = getattr(searcher, "config", None) or {}
    SearchRequest = make_search_request(cfg.get("searchDefaults", {}))

    @app.post("/search", response_model=SearchResponse)
    def search(req: SearchRequest, x_workspace_id: Optional[str] = Header(None)):
        key = None
        if cache is not None:
            payload = req.model_dump()
            payload["_ws"] = get_collection_name(x_workspace_id)
            key = json.dumps(payload, sort_keys=True, ensure_ascii=False)
            hit = cache.get(key)
            if hit is not None:
                return hit
        resp = search_route(searcher, req.query, top_k=req.top_k, symbol_boost=req.symbol_boost,
                            path_prefix=req.path_prefix, kind=req.kind, min_score=req.min_score,
                            x_workspace_id=x_workspace_id)
        if cache is not None:
            cache.set(key, resp, 3600)
        return resp

    @app.get("/search/{symbol}")
    def search_symbol(symbol: str, top_k: int = Query(10, ge=1, le=200), x_workspace_id: Optional[str] = Header(None)):
        rows = searcher.search_by_symbol(symbol, top_k=top_k, collection_name=get_collection_name(x_workspace_id))
        return {"symbol": symbol, "total": len(rows), "results": rows}

    return app
