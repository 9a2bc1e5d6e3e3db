#!/usr/bin/env python3
"""tools/prims_timing.py -- time the library's radix sort and chained scan (ss_prims) against torch's (rocPRIM-backed) sort / cumsum on
the sizes of the reconstruction path (10 M and 19.5 M pairs of 23 / 24-bit keys)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import splashsurf_amd as S  # noqa: E402

L = S.load_library()
L.ss_debug_exclusive_scan_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
L.ss_debug_radix_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_int, C.POINTER(C.c_int), C.c_void_p]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for n, bits in ((10_000_000, 23), (19_544_740, 24), (1_000_000, 21), (4732, 13)):
    k = torch.randint(0, 1 << bits, (n,), device="cuda", dtype=torch.int32)
    k1, v0, v1 = torch.empty_like(k), torch.empty_like(k), torch.empty_like(k)
    res = C.c_int(0)
    kk = k.clone()
    # (the debug entry allocates its work buffer, synchronises and runs on the null stream: its time includes those; the kernels alone are in rocprofv3's trace)
    t_own = timed(lambda: L.ss_debug_radix_sort_pairs(kk.data_ptr(), k1.data_ptr(), v0.data_ptr(), v1.data_ptr(), n, bits, 1, C.byref(res), None))
    t_torch = timed(lambda: torch.sort(k, stable=True))
    x = torch.randint(0, 3, (n,), device="cuda", dtype=torch.int32)
    o = torch.empty_like(x)
    t_scan = timed(lambda: L.ss_debug_exclusive_scan_u32(x.data_ptr(), o.data_ptr(), n, None, None))
    t_cumsum = timed(lambda: torch.cumsum(x, 0))
    print(json.dumps({"n": n, "bits": bits, "sort_own_ms_incl_alloc_sync": round(t_own, 3), "sort_torch_ms": round(t_torch, 3), "scan_own_ms_incl_alloc_sync": round(t_scan, 3),
                      "cumsum_torch_ms": round(t_cumsum, 3)}))
