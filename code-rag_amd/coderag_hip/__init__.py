# -*- coding: utf-8 -*-
"""coderag_hip: MI355X (gfx950) implementation of code-rag's retrieval scoring path.

Host-side mirror of /root/reference/retriever/hybrid_search.py over the C ABI in
include/coderag_hip.h.  No CPU fallback: importing is cheap, but any search needs the in-tree
HIP library and a gfx950 device.
"""
from ._lib import CragError, LIB_PATH, SYMBOLS, load  # noqa: F401
from .collection import (GpuCollection, classify_path, f32_to_bf16_bits, hash_names, merge_rerank_device,  # noqa: F401
                         search_multi, search_multi_device)
from .hybrid_search import ChunkCollection, HybridSearcher, group_requests  # noqa: F401
from .config import load_retrieval_config  # noqa: F401
from .api import create_app, get_collection_name, search_route  # noqa: F401
from .embed import HashingEmbedder  # noqa: F401

__all__ = ["HybridSearcher", "ChunkCollection", "GpuCollection", "CragError", "search_multi", "search_multi_device", "merge_rerank_device",
           "get_collection_name", "search_route", "create_app", "HashingEmbedder", "classify_path", "hash_names",
           "f32_to_bf16_bits", "load", "LIB_PATH", "SYMBOLS", "load_retrieval_config", "group_requests"]
