"""CPU-side checks of the C ABI library: it loads, exports every symbol include/nidx_gpu.h declares,
fails loudly without a device, and its host-only entry points agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from nucliadb_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g

    g.build()
    return _lib.lib()


def test_exports_every_declared_symbol(L):
    header = open(os.path.join(ROOT, "include", "nidx_gpu.h")).read()
    declared = set(re.findall(r"\b(nidx_gpu_[a-z0-9_]+)\s*\(", header))
    declared -= {"nidx_gpu_vector_index_t", "nidx_gpu_bm25_index_t", "nidx_gpu_shard_comm_t"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/nidx_gpu.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.nidx_gpu_abi_version() == 6 == _lib.ABI_VERSION
    assert "#define NIDX_GPU_ABI_VERSION 6" in open(os.path.join(ROOT, "include", "nidx_gpu.h")).read()


def test_no_cpu_fallback_when_device_missing(L):
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = _lib.VectorConfigC(4, 0, 0, 0)
    x = np.zeros((2, 4), np.float32)
    seg = _lib.VectorSegmentC(x.ctypes.data, 16, 2, None, 2, None, 0, 0, None, 0, None, None)
    h = C.c_void_p()
    rc = L.nidx_gpu_vector_open(C.byref(cfg), C.byref(seg), 1, C.byref(h))
    assert rc == _lib.NIDX_ERR_DEVICE and not h
    assert "HIP error" in _lib.last_error()
    out = np.zeros(2, np.float32)
    rc = L.nidx_gpu_similarity(x.ctypes.data, x.ctypes.data, 2, 4, 1, _lib.ORDER_WAVE64, out.ctypes.data)
    assert rc == _lib.NIDX_ERR_DEVICE


_OOM_CHILD = r"""
import ctypes as C, mmap, resource, sys
import numpy as np
from nucliadb_amd import _lib
L = _lib.lib()
n = 1 << 28
scores = mmap.mmap(-1, 4 * n)                       # untouched anonymous zero pages: address space, no RAM
base = C.addressof(C.c_char.from_buffer(scores))
scp, lens, cnt = (C.c_void_p * 1)(base), np.array([n], np.uint32), C.c_uint32()
vm = int(open('/proc/self/statm').read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (vm + (256 << 20), resource.RLIM_INFINITY))
rc = L.nidx_gpu_merge_vector(scp, None, lens.ctypes.data, 1, n, None, None, None, C.byref(cnt))
resource.setrlimit(resource.RLIMIT_AS, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
print(rc, _lib.last_error())
"""


def test_host_allocation_failure_is_an_error_code_not_an_unwind(L):
    # the 2 GiB merge order cannot be allocated under the address-space limit: std::bad_alloc is caught at the boundary
    # (a Rust / cgo caller cannot unwind a C++ exception; before the function-try-blocks this aborted the process)
    import subprocess
    import sys

    r = subprocess.run([sys.executable, "-c", _OOM_CHILD], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == str(_lib.NIDX_ERR_OUT_OF_MEMORY), r.stdout
    assert "host allocation failed" in r.stdout


def test_config_errors(L):
    h = C.c_void_p()
    cfg = _lib.VectorConfigC(0, 0, 0, 0)
    assert L.nidx_gpu_vector_open(C.byref(cfg), None, 0, C.byref(h)) == _lib.NIDX_ERR_INVALID_CONFIGURATION
    assert "dimension cannot be 0" in _lib.last_error()
    cfg = _lib.VectorConfigC(8, 7, 0, 0)
    assert L.nidx_gpu_vector_open(C.byref(cfg), None, 0, C.byref(h)) == _lib.NIDX_ERR_INVALID_CONFIGURATION


def test_use_hnsw_matches_oracle(L, orc):
    rng = np.random.default_rng(5)
    for _ in range(2000):
        total = int(rng.integers(1, 3_000_000))
        matching = int(rng.integers(1, total + 1))
        k = int(rng.integers(1, 200))
        for rq in (0, 1):
            assert bool(L.nidx_gpu_use_hnsw(total, matching, k, rq)) == orc.use_hnsw(total, matching, k, bool(rq))
    # the crossover quoted in SURVEY §8a6: k=10 unfiltered -> HNSW from a few hundred records up
    assert not L.nidx_gpu_use_hnsw(100, 100, 10, 0)
    assert L.nidx_gpu_use_hnsw(100_000, 100_000, 10, 0)


def test_normalize_matches_oracle(L, orc):
    rng = np.random.default_rng(6)
    for d in (3, 10, 64, 758, 768):
        x = rng.normal(size=(5, d)).astype(np.float32)
        out = np.empty_like(x)
        assert L.nidx_gpu_normalize(x.ctypes.data, 5, d, out.ctypes.data) == 0
        for i in range(5):
            assert np.array_equal(out[i].view(np.uint32), orc.normalize(x[i]).view(np.uint32))


def _merge_vector(L, lists, limit):
    n = len(lists)
    sc = [np.array([s for s, _ in l], np.float32) for l in lists]
    ids = [np.array([i for _, i in l], np.uint64) for l in lists]
    lens = np.array([len(l) for l in lists], np.uint32)
    scp = (C.c_void_p * n)(*[a.ctypes.data for a in sc])
    idp = (C.c_void_p * n)(*[a.ctypes.data for a in ids])
    os_, oi, ol = np.zeros(limit, np.float32), np.zeros(limit, np.uint64), np.zeros(limit, np.uint32)
    cnt = C.c_uint32()
    assert L.nidx_gpu_merge_vector(scp, idp, lens.ctypes.data, n, limit, os_.ctypes.data, oi.ctypes.data, ol.ctypes.data, C.byref(cnt)) == 0
    return [(float(os_[i]), int(oi[i])) for i in range(cnt.value)]


def test_merge_vector_matches_oracle(L, orc):
    rng = np.random.default_rng(7)
    for trial in range(200):
        n = int(rng.integers(1, 9))
        lists = []
        for s in range(n):
            m = int(rng.integers(0, 12))
            # coarse scores => many cross-shard ties, the case kmerge's heap order decides
            sc = np.sort(rng.integers(0, 6, m).astype(np.float32) / 4)[::-1]
            lists.append([(float(x), (s << 32) | i) for i, x in enumerate(sc)])
        limit = int(rng.integers(1, 25))
        assert _merge_vector(L, lists, limit) == orc.merge_vector(lists, limit)


def test_merge_bm25_matches_oracle(L, orc):
    rng = np.random.default_rng(8)
    for trial in range(200):
        n = int(rng.integers(1, 9))
        shard_ids = [bytes(rng.integers(97, 100, int(rng.integers(1, 4))).astype(np.uint8)) for _ in range(n)]
        lists = []
        for s in range(n):
            m = int(rng.integers(0, 12))
            sc = np.sort(rng.integers(0, 5, m).astype(np.float32))[::-1]
            addr = np.sort(rng.integers(0, 50, m).astype(np.uint64))
            lists.append([(float(sc[i]), int(addr[i]), shard_ids[s], (s << 16) | i) for i in range(m)])
        limit = int(rng.integers(1, 25))
        want = orc.merge_bm25(lists, limit)
        sc = [np.array([h[0] for h in l], np.float32) for l in lists]
        da = [np.array([h[1] for h in l], np.uint64) for l in lists]
        lens = np.array([len(l) for l in lists], np.uint32)
        sid = [np.frombuffer(b, np.uint8).copy() for b in shard_ids]
        sidl = np.array([len(b) for b in shard_ids], np.uint32)
        P = lambda arrs: (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        os_, od, ol = np.zeros(limit, np.float32), np.zeros(limit, np.uint64), np.zeros(limit, np.uint32)
        cnt = C.c_uint32()
        assert L.nidx_gpu_merge_bm25(P(sc), P(da), lens.ctypes.data, P(sid), sidl.ctypes.data, n, limit, os_.ctypes.data,
                                     od.ctypes.data, ol.ctypes.data, C.byref(cnt)) == 0
        got = [(float(os_[i]), int(od[i]), shard_ids[int(ol[i])]) for i in range(cnt.value)]
        assert got == [(w[0], w[1], w[2]) for w in want], (got, want)


_STRUCTS = {
    "nidx_gpu_vector_config_t": "VectorConfigC", "nidx_gpu_vector_segment_t": "VectorSegmentC",
    "nidx_gpu_vector_search_params_t": "VectorSearchParamsC", "nidx_gpu_filter_index_t": "FilterIndexC",
    "nidx_gpu_filter_op_t": "FilterOpC", "nidx_gpu_filter_program_t": "FilterProgramC", "nidx_gpu_paragraph_t": "ParagraphC",
    "nidx_gpu_segment_dir_contents_t": "SegmentDirContentsC", "nidx_gpu_merge_operand_t": "MergeOperandC",
    "nidx_gpu_bm25_segment_t": "Bm25SegmentC", "nidx_gpu_bm25_clause_t": "Bm25ClauseC",
    "nidx_gpu_bm25_search_after_t": "Bm25SearchAfterC", "nidx_gpu_bm25_search_options_t": "Bm25SearchOptionsC",
    "nidx_gpu_bm25_date_range_t": "Bm25DateRangeC", "nidx_gpu_bm25_prefilter_t": "Bm25PrefilterC",
    "nidx_gpu_ranked_list_t": "RankedListC", "nidx_gpu_facet_count_t": "FacetCountC",
}


def _header_structs():
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "nidx_gpu.h")).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(nidx_gpu_\w+_t)\s*;", h, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            for part in decl.strip().split(","):
                if part.strip():
                    fields.append(re.findall(r"(\w+)\s*(?:\[\d+\])?$", part.strip())[0])
        out[m.group(2)] = fields
    return out


def test_ctypes_structs_have_the_layout_of_the_header(tmp_path):
    """Every struct of include/nidx_gpu.h has a ctypes mirror with the same fields in the same order, the same size and the
    same field offsets as the C compiler gives them (a binding that drifts from the header corrupts arguments silently)."""
    import subprocess

    structs = _header_structs()
    assert set(structs) == set(_STRUCTS), set(structs) ^ set(_STRUCTS)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nidx_gpu.h"', "int main(void) {"]
    for cname, fields in structs.items():
        cls = getattr(_lib, _STRUCTS[cname])
        assert [f[0] for f in cls._fields_] == fields, cname
        lines.append(f'    printf("{cname} %zu", sizeof({cname}));')
        for f in fields:
            lines.append(f'    printf(" %zu", offsetof({cname}, {f}));')
        lines.append('    printf("\\n");')
    lines += ["    return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines():
        cname, size, *offsets = line.split()
        cls = getattr(_lib, _STRUCTS[cname])
        assert C.sizeof(cls) == int(size), cname
        assert [getattr(cls, f[0]).offset for f in cls._fields_] == [int(o) for o in offsets], cname
