#!/bin/bash
# per-phase cycles of the streaming scorer's items (NIDX_GPU_BM25_DEBUG) with and without score floors, bench workload, synchronous entry
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bm25_dbg
mkdir -p $OUT
cd $ROOT
for fl in 1 0; do
  NIDX_GPU_BM25_FLOOR=$fl NIDX_GPU_BM25_DEBUG=1 timeout 600 python - > $OUT/dbg_$fl.log 2>&1 <<'PY'
import sys, types, os
sys.argv=["bench.py","--workload","bm25","--cpu-queries","0"]
import importlib.util
spec=importlib.util.spec_from_file_location("bench","bench.py"); b=importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch
from nucliadb_amd import _lib
a=b.parse(); L=_lib.lib(); dev=torch.device("cuda:0")
bm=b.Bm25Bench(a,L,dev,0,a.n_docs)
for i in range(3): bm.search(i)
PY
  echo "== floor=$fl"; grep "bm25 dbg" $OUT/dbg_$fl.log | tail -8 | cut -c1-400
done
