#!/bin/bash
# A/B of bf16_append's mainloops (NIDX_GPU_BF16_MAINLOOP) by HIP events: parity tests per form, then bench.py --workload bf16 at 4 M and 12.5 M x 1024.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bf16_ab
mkdir -p $OUT
cd $ROOT
rm -f $OUT/test_*.log
for form in ${FORMS:-0 1 2}; do
  if [ ! -f $OUT/test_$form.log ]; then
  NIDX_GPU_BF16_MAINLOOP=$form timeout 600 python -m pytest tests/test_vector_gpu.py -x -q -m gpu -k bf16 > $OUT/test_$form.log 2>&1
  echo "form $form tests: $(tail -1 $OUT/test_$form.log)"
  fi
  for nv in ${NVECS:-4000000 12500000}; do
    NIDX_GPU_BF16_MAINLOOP=$form timeout 600 python bench.py --workload bf16 --n-vectors $nv --dim 1024 --cpu-queries 0 --recall-queries 64 --steps 6 --warmup 2 > $OUT/bench_${form}_$nv.json 2> $OUT/bench_${form}_$nv.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${form}_$nv.json").read().strip().splitlines()[-1])
    print("form $form n=$nv ms_per_step=%.3f frac=%.4f recall=%s flags=%s" % (d["ms_per_step"], d["roofline"]["frac"], d["config"].get("recall_at_10"), d["config"].get("kernel_flags")))
except Exception as e:
    print("form $form n=$nv FAILED", e)
PY
  done
done
