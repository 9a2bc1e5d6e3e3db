"""GPU tests (-m gpu) of the library's own data-parallel primitives (csrc/ss_prims.h / ss_prims.hip): the single-pass chained prefix sum
and the stable LSD radix sort of (u32 key, u32 value) pairs that replace rocPRIM on the reconstruction path.  Checked against
torch.cumsum / a stable torch.sort on the same device arrays: sizes around the tile borders (2048 / 4096), empty input, all-equal keys,
every key width the path uses (1 .. 32 bits), value arrays given and implied (iota), odd and even pass counts."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev():
    """The GPU -- or host memory when the library under test is the CPU execution model of tests/emu (SPLASHSURF_HIP_LIB; its "device" memory is the host's)."""
    import torch
    return "cuda" if torch.cuda.is_available() else "cpu"


def _sync():
    import torch
    if torch.cuda.is_available():
        _sync()


def _lib():
    import splashsurf_amd as S
    L = S.load_library()
    L.ss_debug_exclusive_scan_u32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.ss_debug_radix_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_int, C.POINTER(C.c_int), C.c_void_p]
    return L


@pytest.mark.parametrize("n", [0, 1, 63, 64, 2047, 2048, 2049, 4095, 4096, 4097, 8192, 100_000, 1_000_003, 1_048_575, 1_048_576, 1_056_767, 1_056_769, 20_000_001])
def test_chained_scan_equals_cumsum(n):
    import torch
    L = _lib()
    g = torch.Generator(device=_dev()).manual_seed(n + 1)
    x = torch.randint(0, 50, (max(n, 1),), device=_dev(), dtype=torch.int32, generator=g)[:n]
    out = torch.empty(max(n, 1), device=_dev(), dtype=torch.int32)
    tot = torch.zeros(2, device=_dev(), dtype=torch.int32)
    _sync()
    assert L.ss_debug_exclusive_scan_u32(x.data_ptr(), out.data_ptr(), n, tot.data_ptr(), None) == 0
    ref = torch.cumsum(x.to(torch.int64), 0)
    if n:
        assert torch.equal(out[:n].to(torch.int64), ref - x.to(torch.int64))
        assert int(tot[0]) == int(ref[-1])
    else:
        assert int(tot[0]) == 0


@pytest.mark.parametrize("n,bits,iota", [(0, 8, 1), (1, 8, 0), (100, 3, 1), (4095, 8, 0), (4096, 9, 1), (4097, 16, 0), (50_000, 17, 1), (1_000_003, 23, 1), (1_000_003, 24, 0),
                                         (3_000_001, 25, 1), (3_000_001, 32, 0), (20_000_001, 24, 1)])
def test_radix_sort_is_a_stable_sort(n, bits, iota):
    import torch
    L = _lib()
    g = torch.Generator(device=_dev()).manual_seed(7 * n + bits)
    hi = (1 << bits) if bits < 31 else (1 << 31) - 1
    m = max(n, 1)
    k0 = torch.randint(0, hi, (m,), device=_dev(), dtype=torch.int64, generator=g)
    if bits == 32:
        k0 = k0 * 2 + torch.randint(0, 2, (m,), device=_dev(), dtype=torch.int64, generator=g)
    if n > 1000:
        k0[: n // 3] = k0[0]  # a long run of equal keys: stability
    keys = [k0.to(torch.uint32) if hasattr(torch, "uint32") else None, None]
    ka = (k0 & 0xFFFFFFFF).to(torch.int64)
    buf_k0 = torch.empty(m, device=_dev(), dtype=torch.int32)
    buf_k0.copy_(torch.where(ka >= 2 ** 31, ka - 2 ** 32, ka).to(torch.int32))
    buf_k1 = torch.empty_like(buf_k0)
    v0 = torch.randint(0, 2 ** 31 - 1, (m,), device=_dev(), dtype=torch.int32, generator=g)
    vals = torch.arange(m, device=_dev(), dtype=torch.int32) if iota else v0.clone()
    buf_v0 = torch.full_like(v0, -1) if iota else v0.clone()
    buf_v1 = torch.empty_like(v0)
    res = C.c_int(-1)
    _sync()
    assert L.ss_debug_radix_sort_pairs(buf_k0.data_ptr(), buf_k1.data_ptr(), buf_v0.data_ptr(), buf_v1.data_ptr(), n, bits, iota, C.byref(res), None) == 0
    if n == 0:
        return
    sk, sv = (buf_k0, buf_v0) if res.value == 0 else (buf_k1, buf_v1)
    order = torch.sort(ka[:n], stable=True).indices
    got_k = sk[:n].to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got_k, ka[:n][order])
    assert torch.equal(sv[:n], vals[:n][order])
