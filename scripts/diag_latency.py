"""Latency of a dependent random row fetch (one 4*dim-byte row per step per wave, nothing else in flight) vs footprint."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nucliadb_amd import _lib
L = _lib.lib()
dev = torch.device("cuda", 0)
d = 768
x = torch.rand((10_000_000, d), device=dev, dtype=torch.float32)
torch.cuda.synchronize()
for n in (10_000, 100_000, 1_000_000, 10_000_000):
    for waves, rif in ((4, 1), (4, 2), (1024, 1), (1024, 2), (4096, 2)):
        gpw = 1024
        ms = C.c_float(0)
        _lib.check(L.nidx_gpu_diag_gather(x.data_ptr(), n, d, waves, gpw, rif, 3, C.byref(ms)))
        print("rows=%9d footprint=%6.2f GB waves=%5d rows_in_flight=%d: %.2f us per %d-row step, %.0f GB/s" % (
            n, n * d * 4 / 1e9, waves, rif, ms.value * 1e3 / (gpw / rif), rif, waves * gpw * d * 4 / (ms.value * 1e-3) / 1e9), flush=True)
