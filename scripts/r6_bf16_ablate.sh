#!/bin/bash
# Ablations of bf16_append_ring_kernel (library built with -DNIDX_BF16_ABLATE; results are wrong by construction, only the times matter)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bf16_ablate
mkdir -p $OUT
cd $ROOT
for form in ${FORMS:-2 1}; do
  for abl in ${ABLS:-0 4 1 8 16 32 2 17 49 18 5 21}; do
    NIDX_GPU_BF16_MAINLOOP=$form NIDX_GPU_BF16_ABLATE=$abl timeout 300 python bench.py --workload bf16 --n-vectors ${NVEC:-12500000} --dim 1024 --cpu-queries 0 --recall-queries 0 --steps 5 --warmup 2 > $OUT/b_${form}_$abl.json 2> $OUT/b_${form}_$abl.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/b_${form}_$abl.json").read().strip().splitlines()[-1])
    print("form $form abl %2d ms_per_step=%.3f frac=%.4f flags=%s" % ($abl, d["ms_per_step"], d["roofline"]["frac"], d["config"].get("kernel_flags")))
except Exception as e:
    print("form $form abl $abl FAILED", e)
PY
  done
done
