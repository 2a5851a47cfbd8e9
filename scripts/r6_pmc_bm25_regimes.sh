#!/bin/bash
# SQ instruction counters of bm25_stream_kernel per query regime (scripts/r6_bm25_probe.py): one rocprofv3 --pmc pass per regime.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_bm25_regimes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for regime in ${REGIMES:-bench long1 long1_short2 short3 mid1}; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" ${MORE_SETS:+"SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"}; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_$regime$i -- python $ROOT/scripts/r6_bm25_probe.py --min-s 0.2 $regime > $OUT/log_$regime$i.txt 2>&1
    db=$(ls $OUT/p_$regime$i/*/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python $ROOT/scripts/prof_summary.py $db "regime $regime pass $i: $set" 2>&1 | grep -E "^#|bm25_stream|bm25_merge" > $OUT/summary_${regime}_$i.txt
    rm -rf $OUT/p_$regime$i
  done
done
cat $OUT/summary_*.txt
