"""nucliadb_amd — MI355X-native nidx search hot path (vector k-NN + BM25) behind the reference's interface.

`nucliadb_amd.vector` mirrors `nidx_vector`, `nucliadb_amd.bm25` mirrors the BM25 scoring surface of
`nidx_text` / `nidx_paragraph`, `nucliadb_amd.shard_merge` mirrors `nidx::searcher::shard_merge`.
All of them call libnidx_gpu.so (include/nidx_gpu.h); there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
