"""Golden vectors for the rank-fusion mirror, produced by the REFERENCE's own code.

Run in the build container (the GPU box has no /root/reference):
    python scripts/make_reference_golden.py            # writes tests/golden/rank_fusion_reference.json.gz

The module `nucliadb/src/nucliadb/search/search/rank_fusion.py` is imported from /root/reference unchanged and its
`ReciprocalRankFusion.fuse` / `WeightedCombSum.fuse` are run over seeded random cases (overlapping paragraph ids across the
keyword / semantic / graph lists, tied scores, per-retriever weights, empty and single sources).  Only imports that have nothing
to do with rank fusion are stubbed, because their packages are not in this image: the generated protobuf modules (`nidx_protos`,
`nucliadb_protos`), `nuclia_models`, and `nucliadb.search.search.query_parser` (its `models` is only used by the
`get_rank_fusion` factory, which this script does not call — the algorithms are constructed directly).

What is recorded per case: the algorithm and its parameters, the input lists [(paragraph id, score, score type)] in source
order, and the reference's output [(paragraph id, score as a float64 hex string, score type, [scores the hit carried])].
tests/test_rank_fusion_cpu.py replays the inputs through nucliadb_amd/rank_fusion.py and the native batch routine.
"""
import importlib.abc
import importlib.machinery
import json
import os
import random
import sys
import types
import zlib

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rank_fusion_reference.json.gz")


def install_stubs():
    from pydantic import BaseModel

    class Meta(type):
        def __getattr__(cls, n):
            if n.startswith("__"):
                raise AttributeError(n)
            if n == "Name":
                return lambda v: "V%d" % v
            if n == "Value":
                return lambda s: zlib.crc32(s.encode()) % 100000
            if n.upper() == n:  # an enum member
                return zlib.crc32((cls.__name__ + "." + n).encode()) % 100000
            t = Meta(n, (), {})
            setattr(cls, n, t)
            return t

    class StubModule(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith("__"):
                raise AttributeError(n)
            t = type(n, (BaseModel,), {}) if self.__name__.startswith("nuclia_models") else Meta(n, (), {})
            setattr(self, n, t)
            return t

    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        ROOTS = ("nidx_protos", "nucliadb_protos", "nuclia_models")
        PACKAGES = ("nucliadb.search.search.query_parser",)

        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in self.ROOTS or any(name == p or name.startswith(p + ".") for p in self.PACKAGES):
                return importlib.machinery.ModuleSpec(name, self, is_package=True)
            return None

        def create_module(self, spec):
            m = StubModule(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass

    sys.meta_path.insert(0, Finder())
    sys.path[:0] = [os.path.join(REF, p, "src") for p in ("nucliadb", "nucliadb_models", "nucliadb_utils", "nucliadb_telemetry")]


def main():
    install_stubs()
    import nucliadb.search.search.rank_fusion as rf
    from nucliadb.common.external_index_providers.base import ScoredTextBlock
    from nucliadb.common.ids import ParagraphId
    from nucliadb_models.retrieval import GraphScore, KeywordScore, SemanticScore
    from nucliadb_models.search import SCORE_TYPE

    kinds = {"keyword": (SCORE_TYPE.BM25, KeywordScore), "semantic": (SCORE_TYPE.VECTOR, SemanticScore),
             "graph": (SCORE_TYPE.RELATION_RELEVANCE, GraphScore)}
    import struct

    def f32(x):   # the product's lists carry f32 scores: cases 96.. use values an f32 holds exactly, so the native routine can replay them
        return struct.unpack("f", struct.pack("f", x))[0]

    rng = random.Random(20260924)
    cases = []
    for case in range(144):
        if case == 96:
            rng = random.Random(20260925)   # (cases 0..95 stay what they were)
        exact32 = case >= 96
        algo = "rrf" if case % 2 == 0 and not exact32 else "wcombsum"
        n_ids = rng.choice([3, 8, 20, 60])
        pool = ["%032x/f/file%d/%d-%d" % (rng.randrange(1, 6), rng.randrange(3), 10 * i, 10 * i + 9) for i in range(n_ids)]
        sources = {}
        for name in ("keyword", "semantic", "graph"):
            shape = rng.random()
            n = 0 if shape < 0.15 else rng.randrange(1, min(n_ids, 25) + 1)
            ids = rng.sample(pool, n)
            if name == "graph":
                scores = [1.0] * n  # FAKE_GRAPH_SCORE: every graph hit is tied
            elif rng.random() < 0.3:
                scores = [rng.choice([0.25, 0.5, 0.75, 1.5]) for _ in range(n)]  # heavy ties
            elif name == "keyword":
                scores = [float.fromhex((rng.uniform(0.1, 30.0)).hex()) for _ in range(n)]
            else:
                scores = [rng.uniform(-0.2, 1.0) for _ in range(n)]
            if exact32:
                scores = [f32(x) for x in scores]
            sources[name] = [(i, s) for i, s in zip(ids, scores)]
        weights = {} if rng.random() < 0.4 else {n: rng.choice([0.5, 1.0, 2.0, 3.25]) for n in rng.sample(list(kinds), rng.randrange(1, 4))}
        k = rng.choice([60.0, 2.0, 1.0, 10.5])
        default_weight = rng.choice([1.0, 1.0, 0.7])
        if algo == "rrf":
            fusion = rf.ReciprocalRankFusion(k=k, window=20, weights=weights, default_weight=default_weight)
        else:
            fusion = rf.WeightedCombSum(window=20, weights=weights, default_weight=default_weight)
        inputs = {name: [ScoredTextBlock(paragraph_id=ParagraphId.from_string(pid), score_type=kinds[name][0], scores=[kinds[name][1](score=s)])
                         for pid, s in hits] for name, hits in sources.items()}
        merged = fusion.fuse(inputs)
        cases.append({
            "algorithm": algo, "k": k, "weights": weights, "default_weight": default_weight, "f32_scores": exact32,
            "sources": {name: [[pid, s.hex(), kinds[name][0].value] for pid, s in hits] for name, hits in sources.items()},
            "expected": [[m.paragraph_id.full(), float(m.score).hex(), m.score_type.value, [float(s.score).hex() for s in m.scores]] for m in merged],
        })
    import gzip

    with gzip.GzipFile(OUT, "wb", mtime=0) as gz, __import__("io").TextIOWrapper(gz, encoding="utf-8") as f:
        json.dump({"generated_by": "scripts/make_reference_golden.py", "reference_module": "nucliadb/src/nucliadb/search/search/rank_fusion.py",
                   "cases": cases}, f, separators=(",", ":"))
    print("wrote", OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
