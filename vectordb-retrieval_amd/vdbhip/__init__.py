"""vdbhip -- MI355X-native exact / IVF-Flat k-NN behind the vectordb-retrieval plugin API.

Importing the package does not touch the GPU and does not import torch; the shared library
(libvdbhip.so, hand-written HIP for gfx950) is dlopen'ed on first use and there is no CPU fallback.
"""
from .plugin_api import (ALGORITHM_REGISTRY, INDEXER_REGISTRY, SEARCHER_REGISTRY, BaseAlgorithm, BaseIndexer,
                         BaseSearcher, CompositeAlgorithm, IndexArtifact, get_algorithm_instance, get_indexer_class,
                         get_searcher_class, register_algorithm, register_indexer, register_searcher)
from .algorithms import HipBruteForceIndexer, HipExactSearch, HipLinearSearcher, rerank_candidates
from .index import FlatIndex, merge_packed_partials_device, merge_partials_device
from .ivf import HipApproximateSearch, HipIVFIndexer, HipIVFSearcher, IVFFlatIndex
from . import sharded
from .sharded import HipShardedApproximateSearch, HipShardedExactSearch, shard_bounds

__all__ = [
    "ALGORITHM_REGISTRY", "INDEXER_REGISTRY", "SEARCHER_REGISTRY", "BaseAlgorithm", "BaseIndexer", "BaseSearcher",
    "CompositeAlgorithm", "IndexArtifact", "get_algorithm_instance", "get_indexer_class", "get_searcher_class",
    "register_algorithm", "register_indexer", "register_searcher", "HipExactSearch", "HipBruteForceIndexer",
    "HipLinearSearcher", "rerank_candidates", "FlatIndex", "merge_partials_device", "merge_packed_partials_device",
    "HipApproximateSearch", "HipIVFIndexer", "HipIVFSearcher", "IVFFlatIndex", "HipShardedExactSearch",
    "HipShardedApproximateSearch", "shard_bounds", "sharded",
]
