#!/usr/bin/env python3
"""tools/sort_only.py -- the library's pair sort on 10 M random 24-bit keys, a few times (profiled with rocprofv3 --kernel-trace --stats per variant)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import splashsurf_amd as S
L = S.load_library()
L.ss_debug_radix_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_int, C.POINTER(C.c_int), C.c_void_p]
n, bits = 10_000_000, 24
k = torch.randint(0, 1 << bits, (n,), device="cuda", dtype=torch.int32)
k1, v0, v1 = torch.empty_like(k), torch.empty_like(k), torch.empty_like(k)
res = C.c_int(0)
for _ in range(6):
    kk = k.clone()
    L.ss_debug_radix_sort_pairs(kk.data_ptr(), k1.data_ptr(), v0.data_ptr(), v1.data_ptr(), n, bits, 1, C.byref(res), None)
torch.cuda.synchronize()
print("done")
