#!/usr/bin/env python3
"""gpurun_out/final/summary_{fetch,write}_{corpus}.txt -> profiles/r05_pmc_traffic.json (the HBM bytes per launch bench.py quotes as
roofline.traffic).  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts the 128-B requests of wide coalesced
reads at 64 B, hence x2 (MI355X_MICROARCH.md, HBM section)."""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "final")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
entries = []
for corpus in ("clustered", "uniform"):
    vals = {}
    for p, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(src, "summary_%s_%s.txt" % (p, corpus))
        if not os.path.exists(f):
            continue
        for line in open(f):
            m = re.match(r"void nidx::hnsw_search_kernel<([^>]*)>.*\| %s \| (\d+) \| ([0-9.]+) \|" % counter, line)
            if m:
                vals[counter] = (float(m.group(3)), int(m.group(2)), m.group(1))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        fetch, nd, shape = vals["FETCH_SIZE"]
        write = vals["WRITE_SIZE"][0]
        entries.append({
            "kernel": "hnsw_search_kernel<%s>" % shape.replace(" ", ""),
            "workload": {"corpus": corpus, "n_vectors": n, "dim": 768, "batch": 1024, "k": 10},
            "fetch_size_kib_per_dispatch": fetch, "write_size_kib_per_dispatch": write, "dispatches": nd, "fetch_correction": 2.0,
            "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
            "source": "profiles/r05_pmc_hnsw10m_%s.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace over bench.py "
                      "--corpus %s; FETCH_SIZE x1024 x2 per MI355X_MICROARCH.md HBM section + WRITE_SIZE x1024" % (corpus, corpus)})
# the BM25 scoring kernel on bench.py --workload bm25 (scripts/pmc_bm25_traffic.sh writes gpurun_out/pmc_bm25/traffic_*.txt)
bvals = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(ROOT, "gpurun_out", "pmc_bm25", "traffic_%s.txt" % counter)
    if os.path.exists(f):
        for line in open(f):
            m = re.match(r"void nidx::(bm25_(?:stream|union|fast)_kernel<[^>]*>).*\| %s \| (\d+) \| ([0-9.]+) \|" % counter, line)
            if m:
                bvals[counter] = (float(m.group(3)), int(m.group(2)), m.group(1))
if "FETCH_SIZE" in bvals:
    fetch, nd, kname = bvals["FETCH_SIZE"]
    write = bvals.get("WRITE_SIZE", (0.0, 0, ""))[0]
    entries.append({
        "kernel": kname.replace(" ", ""), "workload": {"corpus": "bm25", "n_vectors": 10_000_000, "dim": 1_000_000, "batch": 1024, "k": 20},
        "fetch_size_kib_per_dispatch": fetch, "write_size_kib_per_dispatch": write, "dispatches": nd, "fetch_correction": 2.0,
        "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
        "source": "profiles/r05_pmc_bm25.txt: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace over bench.py --workload bm25 "
                  "(scripts/pmc_bm25_traffic.sh); FETCH_SIZE x1024 x2 per MI355X_MICROARCH.md HBM section + WRITE_SIZE x1024"})
json.dump({"entries": entries}, open(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(entries, indent=1))
