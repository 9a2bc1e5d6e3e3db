#!/usr/bin/env python3
"""tools/emu_fuzz.py -- randomised parity sweep of the library against the oracle, bit for bit, on whatever build SPLASHSURF_HIP_LIB names: meant for
the CPU execution model of tests/emu (no GPU budget: thousands of cases cost nothing but host time), works on the HIP build as well.

    SPLASHSURF_HIP_LIB=tests/emu/_build/libsplashsurf_emu.so python tools/emu_fuzz.py --cases 400 --seed 1 [--out profiles/r06_emu_fuzz.jsonl]

Unlike tests/test_gpu_fuzz.py (36 + 24 patterned cases) every parameter is drawn independently: cloud kind (jittered lattice, exact lattice, thin
sheet, clusters, BULK fluid block, OVER-DENSE cube -- the last two exercise the certificates and the over-dense path), particle count, length scale,
cube size (cube radius 1 .. 24), smoothing length, subdomain size, threshold, offset, rest density, Real type, strategy, particle AABB, arithmetic
mode, and whether the two-pass splat (certification) is forced.  Every case compares densities, neighbour lists, vertex keys, vertices and triangles
with the oracle exactly like the test does; one JSON line per case, a summary line at the end, exit code 1 on any difference."""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


_ORIG_CLOUD = None


def cloud(rng, n, kind, spacing):
    if kind in ("lattice", "jitter", "sheet", "clusters"):
        return _ORIG_CLOUD(rng, n, kind, spacing)
    if kind == "bulk":  # a box of fluid at rest spacing, slightly jittered: certified sub-blocks inside, a free surface all around
        m = int(round(n ** (1.0 / 3.0))) + 1
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(max(2, m // 2)), indexing="ij"), -1).reshape(-1, 3)
        return (g + rng.random(g.shape) * 0.2) * spacing
    if kind == "dense":  # uniformly random, several times the rest density: blocks with more candidates than a wave's tile
        edge = (n / rng.choice([4.0, 8.0, 12.0])) ** (1.0 / 3.0)
        return rng.random((n, 3)) * edge * spacing
    raise KeyError(kind)


def draw(rng):
    scale = float(rng.choice([1.0, 1.0, 1e-3, 250.0]))
    kind = str(rng.choice(["jitter", "lattice", "sheet", "clusters", "bulk", "bulk", "dense"]))
    c = float(rng.choice([4.0, 2.0, 1.5, 1.3, 1.0, 0.75, 0.5, 0.5, 0.33, 0.2, 0.17]))
    n_max = 6000 if kind in ("bulk", "dense") else 2500
    n = int(rng.integers(40, n_max) * (0.25 if c < 0.25 else (0.5 if c < 0.4 else 1.0))) + 3
    f64 = bool(rng.random() < 0.2)
    strategy = "global" if rng.random() < 0.2 else "grid"
    simd = int(rng.random() < 0.4) if (strategy == "grid" and not f64) else 0
    return dict(seed=int(rng.integers(1 << 30)), n=n, kind=kind, r=0.025 * scale, l=float(rng.choice([2.0, 2.0, 1.5, 2.5])), c=c,
                n_cubes=int(rng.choice([64, 16, 7, 32, 100, 9])), t=float(rng.choice([0.6, 0.6, 0.1, 1.2, 2.0])),
                offset=float(rng.choice([0.0, 0.0, -7.5, 1000.0])) * scale, rest_density=float(rng.choice([1000.0, 1.0, 650.0])), f64=f64,
                strategy=strategy, aabb=bool(rng.random() < 0.15), simd=simd, two_pass=bool(rng.random() < 0.6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-seconds", type=float, default=1e9)
    a = ap.parse_args()
    import test_gpu_fuzz as F
    from oracle import oracle as O
    O.build()
    import splashsurf_amd as S
    from splashsurf_amd.api import Context
    S.load_library()
    ctx, ctx2 = Context(0), Context(0)
    ctx2.set_two_pass(1)
    global _ORIG_CLOUD
    _ORIG_CLOUD = F._cloud
    F._cloud = cloud  # the test's helper builds its cloud with F._cloud: the two extra kinds go through this file's generator
    rng = np.random.default_rng(a.seed)
    out = open(a.out, "w") if a.out else None
    t0 = time.time()
    bad = done = certified = big = 0
    for i in range(a.cases):
        if time.time() - t0 > a.max_seconds:
            break
        case = draw(rng)
        rec = dict(case=i, **case)
        try:
            res = F._run_case(ctx2 if (case["two_pass"] and case["strategy"] == "grid") else ctx, O, case)
            rec.update(ok=True, n_vertices=int(res.mesh.vertices.shape[0]), n_active_blocks=int(res.stats["n_active_blocks"]),
                       n_certified_subblocks=int(res.stats["n_certified_subblocks"]), n_large_tile_blocks=int(res.stats["n_large_tile_blocks"]))
            certified += rec["n_certified_subblocks"] > 0
            big += rec["n_large_tile_blocks"] > 0
        except AssertionError:
            rec.update(ok=False, error=traceback.format_exc(limit=3)[-600:])
            bad += 1
        except Exception as e:  # grid construction errors of the library must be the oracle's too: _run_case lets them through as they come
            rec.update(ok=False, error=repr(e)[:600])
            bad += 1
        done += 1
        line = json.dumps(rec)
        if out:
            out.write(line + "\n")
            out.flush()
        if not rec["ok"] or i % 25 == 0:
            print(line, flush=True)
    summary = dict(summary=True, library=os.environ.get("SPLASHSURF_HIP_LIB", "splashsurf_amd/libsplashsurf_hip.so"), seed=a.seed, cases=done, failed=bad,
                   cases_with_certified_subblocks=certified, cases_with_over_dense_blocks=big, seconds=round(time.time() - t0, 1))
    print(json.dumps(summary), flush=True)
    if out:
        out.write(json.dumps(summary) + "\n")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
