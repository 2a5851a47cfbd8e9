#!/bin/bash
# ASan + UBSan fuzz of the host-side parsers of untrusted files (segment directories, hnsw.graph).  CPU only, ~1 minute.
#   scripts/fuzz/run.sh [iterations]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
WORK=$(mktemp -d /tmp/nidx_fuzz.XXXXXX)
ITER=${1:-4000}
CXXFLAGS="-std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/nucliadb_amd/csrc"
g++ $CXXFLAGS "$ROOT/scripts/fuzz/fuzz_host_parsers.cpp" "$ROOT/scripts/fuzz/stub_errors.cpp" \
    "$ROOT/nucliadb_amd/csrc/segment_dir.cpp" "$ROOT/nucliadb_amd/csrc/segment_v1.cpp" "$ROOT/nucliadb_amd/csrc/fst_index.cpp" "$ROOT/nucliadb_amd/csrc/hnsw_graph.cpp" -o "$WORK/fuzz"
mkdir -p "$WORK/seed" "$WORK/seed_v1" "$WORK/scratch"
cd "$ROOT"
python - "$WORK/seed" <<'PY'
import sys, uuid
import numpy as np
from oracle import oracle as orc
from nucliadb_amd.vector import VectorSegment
orc.build()
rng = np.random.default_rng(7)
n, d = 300, 64
x = rng.standard_normal((n, d)).astype(np.float32)
x /= np.linalg.norm(x, axis=1, keepdims=True)
seg = orc.Segment(x, similarity=orc.SIM_DOT)
graph, edges = seg.build_graph(2).serialize_v2(n)
rids = [str(uuid.UUID(int=0x7000 + i)) for i in range(9)]
keys = [f"{rids[i % 9]}/{'t/title' if i % 3 else 'a/body2'}/{i}-{i + 3}" for i in range(n)]
labels = [[f"/l/set/{i % 5}"] * (i % 3 != 0) + ["/e/x/y"] * (i % 11 == 0) for i in range(n)]
meta = [bytes(rng.integers(0, 256, i % 9, dtype=np.uint8)) for i in range(n)]
quant = rng.integers(0, 256, (n, d // 8 + 8), dtype=np.uint8)
VectorSegment(keys, x, labels, meta, graph=graph.tobytes(), graph_edges=edges, quantized=quant).save(sys.argv[1])
# the same segment in the pre-migration formats (nodes.kv + index.hnsw)
v1 = sys.argv[1] + "_v1"
open(v1 + "/nodes.kv", "wb").write(orc.nodes_kv_bytes(d, x, keys, labels, meta))
layers, entry = orc.parse_hnsw_v2(graph.tobytes(), edges, n)
open(v1 + "/index.hnsw", "wb").write(orc.disk_hnsw_v1_bytes(n, layers, entry))
PY
ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 "$WORK/fuzz" "$WORK/seed" "$WORK/scratch" 64 "$ITER" "$WORK/seed_v1"
rm -rf "$WORK"
