#!/bin/bash
# PMC passes over the bf16 scan (12.5M x 1024, batch 1024): where the waves' cycles go and the L2 hit rate.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_bf16
mkdir -p $OUT
BENCH="python $ROOT/bench.py --workload bf16 --n-vectors ${NVEC:-12500000} --dim 1024 --cpu-queries 0 --clustered-n 0 --steps 2 --warmup 1 --recall-queries 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $OUT/sq -- $BENCH > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $OUT/tcc -- $BENCH > /dev/null 2>&1
python3 - <<PY
import glob, sqlite3
for sub in ("sq", "tcc"):
    for db in glob.glob("$OUT/%s/**/*.db" % sub, recursive=True):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if "pmc_event" in t][0]
        info = [t for t in tabs if "info_pmc" in t][0]
        disp = [t for t in tabs if "kernel_dispatch" in t][0]
        sym = [t for t in tabs if "kernel_symbol" in t][0]
        q = f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id=i.id join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id where s.kernel_name like '%bf16_scan%' group by 1,2"
        try:
            for r in c.execute(q):
                print(sub, r[1], r[2], "dispatches", r[3])
        except Exception as ex:
            print("query failed", ex, tabs)
PY
