#!/bin/bash
# BM25 scoring kernel vs batch size under rocprofv3 --kernel-trace --stats (GPU box).  Prints one line per batch.
cd /tmp && export TMPDIR=/tmp
for b in ${@:-256 1024 4096 16384}; do
  rm -rf /tmp/prof_$b
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$b -- python $GRAFT_REPO_ROOT/bench.py --workload bm25 --batch $b --steps 4 --warmup 1 --cpu-queries 0 > /tmp/out_$b.json 2>/dev/null
  python - $b <<'P'
import json, sys, glob, subprocess, os
b = sys.argv[1]
d = json.load(open("/tmp/out_%s.json" % b))
db = sorted(glob.glob("/tmp/prof_%s/*/*.db" % b))[0]
out = subprocess.run([sys.executable, os.environ["GRAFT_REPO_ROOT"] + "/scripts/prof_summary.py", db, "x"], capture_output=True, text=True).stdout
fast = [l for l in out.splitlines() if "bm25_stream_kernel" in l or "bm25_union_kernel" in l or "bm25_fast_kernel" in l][0].split("|")
alone = (d["roofline"].get("scoring_alone") or {})
print("%-6s | postings/batch %.0f | scoring + merge launch, avg us under rocprofv3 (launches of several threads overlap) %s | one launch at a time by HIP events us %.1f = %.3f of HBM | the scoring kernel alone (two-launch mode) us %.1f = %.3f | end-to-end %.1f G postings/s" % (
    b, d["config"]["postings_per_batch"], fast[3].strip(), d["roofline"]["kernel_ms"] * 1000, d["roofline"]["frac"], (alone.get("kernel_ms") or 0) * 1000, alone.get("frac") or 0, d["value"] / 1e9))
P
done
