"""Host-side mirror of the reference's plugin surface for the brute-force / IVF-Flat path.

The reference cannot be imported next to this package (its `src.algorithms` pulls in faiss), so the
surface is re-declared here with identical names, argument meaning and error behaviour:

  BaseAlgorithm(name, dimension, **kwargs)           <- src/algorithms/base_algorithm.py:5-123
      build_index / search / batch_search / get_name / get_parameters /
      record_operation / get_operations / save_index / load_index
  IndexArtifact, BaseIndexer, BaseSearcher            <- src/algorithms/modular.py:19-82
  INDEXER_REGISTRY / SEARCHER_REGISTRY + register_* / get_*_class   <- modular.py:85-106
  CompositeAlgorithm                                   <- modular.py:554-622
  ALGORITHM_REGISTRY + get_algorithm_instance          <- src/algorithms/__init__.py:25-47

A maintainer of the reference drops the Hip* classes of `algorithms.py` into these registries
unchanged (INTEGRATION.md shows the three-line registration).
"""
from __future__ import annotations

import copy
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Type

import numpy as np

Metadata = Optional[List[Dict[str, Any]]]
SearchResult = Tuple[np.ndarray, np.ndarray]


class BaseAlgorithm(ABC):
    """Plugin contract the experiment loop drives (experiment_runner.py:320-455)."""

    def __init__(self, name: str, dimension: int, **kwargs: Any) -> None:
        self.name = name
        self.dimension = dimension
        self.config = kwargs              # serialised into <name>_results.json by the harness
        self.vectors = None
        self.metadata = None
        self.index_built = False
        self.build_time = -1.0
        self.index_memory_usage = -1.0
        self.operation_counter: Dict[str, Any] = {}

    @abstractmethod
    def build_index(self, vectors: np.ndarray, metadata: Metadata = None) -> None:
        """Index `vectors` (n, dimension)."""

    @abstractmethod
    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        """k nearest neighbours of one query -> (distances (k,), indices (k,))."""

    @abstractmethod
    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        """k nearest neighbours of each query -> (distances (n,k), indices (n,k))."""

    def get_name(self) -> str:
        return self.name

    def get_parameters(self) -> Dict[str, Any]:
        return self.config

    def record_operation(self, key: str, value: float) -> None:
        self.operation_counter[key] = float(self.operation_counter.get(key, 0.0)) + float(value)

    def get_operations(self) -> Dict[str, Any]:
        return dict(self.operation_counter)

    def save_index(self, artifact_dir: str, context: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        raise NotImplementedError(f"{type(self).__name__} does not support index persistence")

    def load_index(self, artifact_dir: str, context: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        raise NotImplementedError(f"{type(self).__name__} does not support index persistence")

    def __str__(self) -> str:
        return f"{self.name} (dimension={self.dimension}, parameters={self.config})"


@dataclass
class IndexArtifact:
    """What an indexer hands to a searcher: a kind tag, the payload and free-form metadata."""

    kind: str
    data: Any
    metadata: Dict[str, Any] = field(default_factory=dict)


class _Component(ABC):
    def __init__(self, name: str, dimension: int, metric: str = "l2", **kwargs: Any) -> None:
        self.name = name
        self.dimension = dimension
        self.metric = metric
        self.params = kwargs

    def describe(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {"name": self.name, "type": type(self).__name__, "metric": self.metric}
        if self.params:
            out["params"] = copy.deepcopy(self.params)
        return out


class BaseIndexer(_Component):
    @abstractmethod
    def build(self, vectors: np.ndarray, metadata: Metadata = None) -> IndexArtifact:
        """Build an index artifact from `vectors`."""


class BaseSearcher(_Component):
    def __init__(self, name: str, dimension: int, metric: str = "l2", **kwargs: Any) -> None:
        super().__init__(name, dimension, metric, **kwargs)
        self._prepared = False

    @abstractmethod
    def attach(self, artifact: IndexArtifact, vectors: np.ndarray, metadata: Metadata = None) -> None:
        """Bind to an artifact before serving queries."""

    @abstractmethod
    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        ...

    @abstractmethod
    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        ...


INDEXER_REGISTRY: Dict[str, Type[BaseIndexer]] = {}
SEARCHER_REGISTRY: Dict[str, Type[BaseSearcher]] = {}
ALGORITHM_REGISTRY: Dict[str, Type[BaseAlgorithm]] = {}


def register_indexer(name: str, cls: Type[BaseIndexer]) -> None:
    INDEXER_REGISTRY[name] = cls


def register_searcher(name: str, cls: Type[BaseSearcher]) -> None:
    SEARCHER_REGISTRY[name] = cls


def register_algorithm(name: str, cls: Type[BaseAlgorithm]) -> None:
    ALGORITHM_REGISTRY[name] = cls


def get_indexer_class(name: str) -> Type[BaseIndexer]:
    try:
        return INDEXER_REGISTRY[name]
    except KeyError:
        raise ValueError(f"Unknown indexer type '{name}'. Available: {list(INDEXER_REGISTRY)}") from None


def get_searcher_class(name: str) -> Type[BaseSearcher]:
    try:
        return SEARCHER_REGISTRY[name]
    except KeyError:
        raise ValueError(f"Unknown searcher type '{name}'. Available: {list(SEARCHER_REGISTRY)}") from None


class CompositeAlgorithm(BaseAlgorithm):
    """Indexer + searcher pair behind the BaseAlgorithm contract (YAML `indexer_ref` / `searcher_ref`)."""

    def __init__(self, name: str, dimension: int, indexer: Dict[str, Any], searcher: Dict[str, Any],
                 metric: str = "l2", **kwargs: Any) -> None:
        super().__init__(name, dimension)
        if not indexer or not searcher:
            raise ValueError("Both indexer_config and searcher_config must be provided for CompositeAlgorithm")
        self.metric = metric
        self.extra_params = kwargs
        self.index_artifact: Optional[IndexArtifact] = None
        self.indexer_config = copy.deepcopy(indexer)
        self.searcher_config = copy.deepcopy(searcher)
        self.indexer = self._make(self.indexer_config, get_indexer_class, "Indexer")
        self.searcher = self._make(self.searcher_config, get_searcher_class, "Searcher")
        self.config = {"metric": metric, "indexer": self.indexer.describe(), "searcher": self.searcher.describe()}
        if kwargs:
            self.config["params"] = copy.deepcopy(kwargs)

    def _make(self, cfg: Dict[str, Any], lookup, what: str):
        cfg = copy.deepcopy(cfg)
        kind = cfg.pop("type", None)
        if kind is None:
            raise ValueError(f"{what} configuration must include a 'type' field")
        return lookup(kind)(name=cfg.pop("name", kind), dimension=self.dimension,
                            metric=cfg.pop("metric", self.metric), **cfg)

    def build_index(self, vectors: np.ndarray, metadata: Metadata = None) -> None:
        self.index_artifact = self.indexer.build(vectors, metadata)
        self.searcher.attach(self.index_artifact, vectors, metadata)
        self.index_built = True

    def _require_built(self) -> None:
        if not self.index_built:
            raise RuntimeError("Index has not been built for this algorithm")

    def search(self, query: np.ndarray, k: int = 10) -> SearchResult:
        self._require_built()
        return self.searcher.search(query, k)

    def batch_search(self, queries: np.ndarray, k: int = 10) -> SearchResult:
        self._require_built()
        return self.searcher.batch_search(queries, k)

    def get_memory_usage(self):
        fn = getattr(self.searcher, "get_memory_usage", None)
        return fn() if fn else None


for _alias in ("Composite", "CompositeAlgorithm", "Modular"):
    register_algorithm(_alias, CompositeAlgorithm)


def get_algorithm_instance(algorithm_type: str, dimension: int, **params: Any) -> BaseAlgorithm:
    """Factory keyed by the YAML `type:` string (src/algorithms/__init__.py:37-47)."""
    if algorithm_type not in ALGORITHM_REGISTRY:
        raise ValueError(f"Unknown algorithm type: {algorithm_type}. Available types: {list(ALGORITHM_REGISTRY)}")
    name = params.pop("name", algorithm_type)
    return ALGORITHM_REGISTRY[algorithm_type](name=name, dimension=dimension, **params)
