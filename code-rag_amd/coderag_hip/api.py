# -*- coding: utf-8 -*-
"""The `/search` route's logic without the web framework:
collection naming (api/main.py:77-82), SearchResult mapping and the three post-filters applied
AFTER top_k truncation (api/main.py:191-216).  Response shape: {query, total, results:[{id,
score, name, kind, path, start_line, end_line, text_preview}]} (api/main.py:151-164).
The Redis cache of api/cache.py is deliberately absent: it bypasses the scoring path."""
from __future__ import annotations

import re
from typing import Any, Dict, Optional


def get_collection_name(workspace_id: Optional[str]) -> str:
    raw = (workspace_id or "").strip()
    if not raw or raw == "default":
        return "code_chunks"
    safe_id = re.sub(r"[^a-zA-Z0-9_-]", "_", raw)[:64] or "default"
    return "code_chunks" if safe_id == "default" else f"code_chunks__{safe_id}"


def search_route(searcher, query: str, top_k: int = 10, symbol_boost: float = 2.0, path_prefix: Optional[str] = None,
                 kind: Optional[str] = None, min_score: Optional[float] = None,
                 x_workspace_id: Optional[str] = None) -> Dict[str, Any]:
    col_name = get_collection_name(x_workspace_id)
    rs = searcher.search(query, top_k=top_k, symbol_boost=symbol_boost, include_documents=False,
                         collection_name=col_name)
    out = []
    for r in rs:
        m = r["metadata"]
        out.append({
            "id": r["id"],
            "score": float(r["score"]),
            "name": str(m.get("name", "")),
            "kind": str(m.get("kind", "")),
            "path": str(m.get("path", "")),
            "start_line": int(m.get("start_line", 0) or 0),
            "end_line": int(m.get("end_line", 0) or 0),
            "text_preview": str(r.get("text_preview", "")),
        })
    if path_prefix:
        pref = path_prefix.replace("\\", "/")
        out = [x for x in out if x["path"].replace("\\", "/").startswith(pref)]
    if kind:
        out = [x for x in out if (x["kind"] or "").lower() == kind.lower()]
    if min_score is not None:
        out = [x for x in out if x["score"] >= float(min_score)]
    return {"query": query, "total": len(out), "results": out}


def create_app(searcher, cache=None):
    """The two retrieval routes of api/main.py as a FastAPI application around `searcher` (http_app.py).
    fastapi is imported on this call, not at package import: the scoring path does not depend on it."""
    from .http_app import build_app
    return build_app(searcher, cache)
