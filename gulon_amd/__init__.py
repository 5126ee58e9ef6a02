"""gulon_amd: MI355X-native ANN index build + query path behind Gulon's
Index / ProductQuantizer / KMeans API (host mirror over libgulon_hip.so)."""
from . import native
from .coder import Coder, width_for_clusters
from .index import Index, PQIndex, Result, SortedIndex, exact_nearest_neighbours, prepare_query, tune_live
from .grouped import GroupedIndex, GroupedVectors, LimitGroups, LimitVectors, group
from .kmeans import KMeans
from .kmeans import Config as KMeansConfig
from .matrix import DeviceMatrix, Matrix
from .product_quantizer import EncodedMatrix, ProductQuantizer, Quantizer
from .product_quantizer import Config as ProductQuantizerConfig
from .vectors import Vectors, subvector_bounds, subvectors
from .word_vectors import (GroupedWordVectors, KeyedIndex, KeyIndexGrouped, KeyIndexSorted, WordVectors,
                           read_word2vec)

__all__ = ["native", "GroupedIndex", "GroupedVectors", "LimitGroups", "LimitVectors", "group", "Coder", "width_for_clusters", "Index", "PQIndex", "Result", "SortedIndex",
           "exact_nearest_neighbours", "prepare_query", "tune_live", "KMeans", "KMeansConfig", "DeviceMatrix", "Matrix",
           "EncodedMatrix", "ProductQuantizer", "Quantizer", "ProductQuantizerConfig", "Vectors",
           "subvector_bounds", "subvectors", "GroupedWordVectors", "KeyedIndex", "KeyIndexGrouped", "KeyIndexSorted",
           "WordVectors", "read_word2vec"]
