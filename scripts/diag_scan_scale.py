"""Diagnostic: exact-scan routes vs the oracle on the clustered corpus at growing n."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nucliadb_amd import _lib
from oracle import oracle as orc

orc.build()
L = _lib.lib()
dev = torch.device("cuda", 0)
D, K = 768, 10
kind = sys.argv[1] if len(sys.argv) > 1 else "clustered"
for n in [int(v) for v in (sys.argv[2:] or ["100000", "400000", "1000000"])]:
    for B in (256, 1024):
        x = bench.gen_corpus(kind, n, D, dev, 1234567890)
        q = bench.gen_queries(kind, x, 1, B, D, dev, 2)[0].contiguous()
        cfg = _lib.VectorConfigC(D, 1, 0, 0)
        cseg = _lib.VectorSegmentC(x.data_ptr(), D * 4, n, None, n, None, 0, 0, None, 0, None, None)
        h = C.c_void_p()
        _lib.check(L.nidx_gpu_vector_open(C.byref(cfg), C.byref(cseg), 1, C.byref(h)))
        xh = x.cpu().numpy()
        del x
        ov = torch.zeros((B, K), dtype=torch.int32, device=dev)
        os_ = torch.zeros((B, K), dtype=torch.float32, device=dev)
        oc = torch.zeros((B,), dtype=torch.int32, device=dev)
        res = {}
        for name, env in (("shared", "1"), ("tile", "0")):
            os.environ["NIDX_GPU_SCAN_SHARED"] = env
            p = _lib.VectorSearchParamsC(K, -1.0, 1, _lib.METHOD_BRUTE_FORCE)
            _lib.check(L.nidx_gpu_vector_segment_search_device(h, 0, q.data_ptr(), B, C.byref(p), None, ov.data_ptr(), os_.data_ptr(), oc.data_ptr(),
                                                               None, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            res[name] = (ov.cpu().numpy().view(np.uint32).copy(), os_.cpu().numpy().copy())
        oseg = orc.Segment(xh, similarity=orc.SIM_COSINE, order=orc.ORDER_WAVE64)
        bv, bs, bc = oseg.brute_force_batch(q.cpu().numpy()[:8], K, threads=8)
        for name in res:
            same = sum(np.array_equal(bv[i], res[name][0][i]) for i in range(8))
            print(kind, n, B, name, "ids equal to oracle: %d/8" % same, flush=True)
        same = sum(np.array_equal(res["shared"][0][i], res["tile"][0][i]) for i in range(B))
        print(kind, n, B, "shared == tile: %d/%d" % (same, B), "row0 shared", res["shared"][0][0][:5], res["shared"][1][0][:3], "tile", res["tile"][0][0][:5], res["tile"][1][0][:3],
              "oracle", bv[0][:5], bs[0][:3], flush=True)
        L.nidx_gpu_vector_close(h)
