"""tools/emu_schedule_digest.py -- one digest over five reconstructions (fine / coarse grid, both arithmetics, certification forced, an over-dense cube); run it on the emulated
library under different HIP_EMU_SHUFFLE seeds / HIP_EMU_THREADS: the digest must not move (profiles/r06_emu_schedule_shuffle.txt)."""
import sys, hashlib, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import splashsurf_amd as S
from splashsurf_amd import workloads as W
from splashsurf_amd.api import Context
S.load_library()
c2 = Context(0); c2.set_two_pass(1)
pts = W.tank_particles(0.12)
h = hashlib.sha256()
for simd in (False, True):
    for cube in (0.5, 1.5):
        r = S.reconstruct_surface(pts, context=c2, particle_radius=0.005, smoothing_length=2.0, cube_size=cube, subdomain_grid_auto_disable=False, simd=simd)
        h.update(r.mesh.vertices.tobytes()); h.update(r.mesh.triangles_u32.tobytes()); h.update(r.particle_densities.tobytes()); h.update(r.vertex_keys.tobytes())
pts = np.random.default_rng(5).random((40000, 3)).astype(np.float32) * 0.25   # over-dense
r = S.reconstruct_surface(pts, context=c2, particle_radius=0.005, smoothing_length=2.0, cube_size=0.5, subdomain_grid_auto_disable=False, simd=False)
h.update(r.mesh.vertices.tobytes()); h.update(r.mesh.triangles_u32.tobytes()); h.update(r.particle_densities.tobytes())
print(h.hexdigest()[:24], r.stats["n_large_tile_blocks"])
