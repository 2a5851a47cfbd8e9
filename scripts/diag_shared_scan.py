import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from test_vector_gpu import gpu_search, unit_rows, bits
from nucliadb_amd import _lib
n, d, nq, k, sim = 20003, 768, 200, 10, 1
rng = np.random.default_rng(n + d + sim)
x = unit_rows(rng, n, d)
x[200:230] = x[11]
q = rng.normal(size=(nq, d)).astype(np.float32)
q[0] = x[11]
os.environ["NIDX_GPU_SCAN_SHARED"] = "1"
ov, osc, oc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE)
os.environ["NIDX_GPU_SCAN_SHARED"] = "0"
rv, rsc, rc = gpu_search(x, sim, q, k, method=_lib.METHOD_BRUTE_FORCE)
bad = np.argwhere((ov != rv) | (bits(osc) != bits(rsc)))
print("mismatches", len(bad))
for i, j in bad[:12]:
    print(i, j, ov[i, j], rv[i, j], osc[i, j], rsc[i, j], "row%8", ov[i, j] % 8, rv[i, j] % 8)
