"""Synthetic particle workloads of BASELINE.json / SURVEY.md section 8(d) (seeded, reproducible).

Shared by bench.py, the parity tests and tools/gen_goldens.py.  Pure numpy; no GPU needed.
"""
import numpy as np


def uniform_cube_particles(n, seed=12345):
    """S1M / S10M-cube: `n` uniform-random float32 points in [0,1)^3 (config 2: n=1e6, seed 12345)."""
    return np.random.default_rng(seed).random((int(n), 3), dtype=np.float32)


def _jittered_block(nx, ny, nz, spacing, origin, seed):
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    jitter = (np.random.default_rng(seed).random(g.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.5)
    return ((g + np.float32(0.5) + jitter) * np.float32(spacing) + np.asarray(origin, dtype=np.float32)).astype(np.float32)


def tank_particles(scale=1.0, particle_radius=0.005):
    """S10M-tank (config 3) and scaled variants: two jittered-lattice fluid blocks at rest spacing 2r.

    scale=1: blocks of 125x200x200 sites (N = 10 000 000): A = [0,1.25)x[0,2)x[0,2),
    B = [2.75,4)x[0,2)x[2,4); jitter U(-0.25,0.25)*spacing, seeds 3 (A) and 4 (B).
    Other scales multiply the site counts and the block-B origin (scale = 4**(1/3) gives S40M-tank).
    """
    s = 2.0 * particle_radius
    nx, ny, nz = max(1, round(125 * scale)), max(1, round(200 * scale)), max(1, round(200 * scale))
    a = _jittered_block(nx, ny, nz, s, (0.0, 0.0, 0.0), 3)
    b = _jittered_block(nx, ny, nz, s, (2.75 * scale, 0.0, 2.0 * scale), 4)
    return np.concatenate([a, b], axis=0)


def tank_slab_particles(rank, world, scale=1.0, particle_radius=0.005):
    """Weak-scaling variant for multi-GPU runs: rank `rank` of `world` gets its own copy of the tank,
    translated along y by rank * tank height so that the union is one tall fluid column."""
    p = tank_particles(scale, particle_radius)
    # tanks are stacked seamlessly (height = ny * spacing), so neighbouring ranks share real halos
    height = max(1, round(200 * scale)) * 2.0 * particle_radius
    p[:, 1] += np.float32(rank * height)
    return p


def _data_file(name):
    import os
    return np.ascontiguousarray(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "data", name)), dtype=np.float32)


WORKLOADS = {
    # name: (generator kwargs, particle_radius, smoothing_length, cube_size) -- radius-relative l and c
    "s1m": dict(gen=lambda: uniform_cube_particles(1_000_000, 12345), particle_radius=0.01, smoothing_length=2.0, cube_size=1.0),
    "s10m_tank": dict(gen=lambda: tank_particles(1.0), particle_radius=0.005, smoothing_length=2.0, cube_size=0.5),
    "s10m_cube": dict(gen=lambda: uniform_cube_particles(10_000_000, 12346), particle_radius=0.005, smoothing_length=2.0, cube_size=0.5),
    "s40m_tank": dict(gen=lambda: tank_particles(4.0 ** (1.0 / 3.0)), particle_radius=0.005, smoothing_length=2.0, cube_size=0.5),
    "tank_small": dict(gen=lambda: tank_particles(0.08), particle_radius=0.005, smoothing_length=2.0, cube_size=0.5),
    # the two data-file configurations of BASELINE.json (configs[0] and configs[4])
    "config1": dict(gen=lambda: _data_file("double_dam_break_frame_26_4732_particles.npy"), particle_radius=0.025, smoothing_length=2.0, cube_size=1.1),
    "config5": dict(gen=lambda: _data_file("hilbert_46843_particles.npy"), particle_radius=0.025, smoothing_length=2.0, cube_size=0.45),
}
