# -*- coding: utf-8 -*-
"""configs/retrieval.yaml: the retrieval knobs north_star names (topK, symbolBoost).

In the reference that file is 0 bytes and nothing reads YAML; the knobs live in the VS Code
extension's settings rag.topK (8) / rag.symbolBoost (2) (clients/vscode/package.json:75-84, sent with every
request by clients/vscode/src/api.ts:80-81) and in the request-model defaults of the /search route
(api/main.py:145-146).  Here the file is real and this module is its reader:

  * `topK` / `symbolBoost`  -> what HybridSearcher.search uses when the caller passes None for top_k /
                               symbol_boost (the signature's own defaults, 10 and 2.0, are the reference's and stay)
  * `searchDefaults`        -> defaults of the /search request model (coderag_hip/http_app.py)
  * `metric`, `dtype`       -> geometry of collections this searcher creates
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

DEFAULTS: Dict[str, Any] = {
    "topK": 8,
    "symbolBoost": 2.0,
    "searchDefaults": {"top_k": 10, "symbol_boost": 2.0},
    "metric": "l2",
    "dtype": "f32",
}

DEFAULT_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "configs",
                            "retrieval.yaml")


def load_retrieval_config(path: Optional[str] = None) -> Dict[str, Any]:
    """The knobs as a dict, DEFAULTS filled in for anything the file leaves out.  `path` None reads
    $CODERAG_RETRIEVAL_CONFIG or this repository's configs/retrieval.yaml; a missing or empty file (the reference's
    state) gives the defaults."""
    path = path or os.environ.get("CODERAG_RETRIEVAL_CONFIG") or DEFAULT_PATH
    cfg: Dict[str, Any] = {k: (dict(v) if isinstance(v, dict) else v) for k, v in DEFAULTS.items()}
    if os.path.exists(path) and os.path.getsize(path) > 0:
        import yaml
        with open(path, "r", encoding="utf-8") as f:
            data = yaml.safe_load(f) or {}
        if not isinstance(data, dict):
            raise ValueError(f"{path}: expected a mapping at the top level")
        for k, v in data.items():
            if k == "searchDefaults" and isinstance(v, dict):
                cfg["searchDefaults"].update(v)
            else:
                cfg[k] = v
    top_k, sb = int(cfg["topK"]), float(cfg["symbolBoost"])
    if not 1 <= top_k <= 200:
        raise ValueError(f"{path}: topK must be in 1..200 (CRAG_MAX_TOP_K), got {top_k}")
    if cfg["metric"] not in ("l2", "ip") or cfg["dtype"] not in ("f32", "bf16"):
        raise ValueError(f"{path}: metric must be l2|ip and dtype f32|bf16")
    cfg["topK"], cfg["symbolBoost"] = top_k, sb
    cfg["searchDefaults"]["top_k"] = int(cfg["searchDefaults"]["top_k"])
    cfg["searchDefaults"]["symbol_boost"] = float(cfg["searchDefaults"]["symbol_boost"])
    return cfg
