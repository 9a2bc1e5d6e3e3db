"""CPU tests (-m "not gpu"): the oracle against the reference's golden vectors and known answers.

The goldens under tests/golden/ were produced by the reference itself (tools/gen_goldens.py); these
tests keep the oracle pinned to them on every run.
"""
import hashlib

import numpy as np
import pytest

import mesh_compare as MC
from conftest import golden_input, golden_params, load_golden

FULL = ["kat1", "edge_empty", "edge_single", "edge_coincident", "edge_aabb_excludes_all", "cube_2366_aabb", "cube_8",
        "free_particles_125", "cube_2366", "bunny_7705", "config1_double_dam_break", "cube_2366_n16"]
DIGEST = ["config5_hilbert", "tank_small"]


def run_oracle(O, pts, prm):
    kw = {}
    if "aabb_min" in prm:
        kw = dict(aabb_min=np.asarray(prm["aabb_min"], np.float32), aabb_max=np.asarray(prm["aabb_max"], np.float32))
    par = O.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"],
                                 iso_surface_threshold=prm["iso_surface_threshold"],
                                 subdomain_num_cubes_per_dim=prm.get("subdomain_num_cubes_per_dim", 64), **kw)
    return O.reconstruct_surface(pts, par)


@pytest.mark.parametrize("name", FULL)
def test_oracle_matches_reference_full(oracle, name):
    g = load_golden(name)
    pts = golden_input(g)
    res = run_oracle(oracle, pts, golden_params(g))
    assert np.array_equal(res.grid["n_cells"], g["n_cells"])
    assert np.array_equal(res.grid["aabb_min"].view(np.uint32), g["grid_min"].view(np.uint32))
    # densities: bit-identical to the reference
    assert np.array_equal(res.particle_densities.view(np.uint32), g["densities"].view(np.uint32))
    if "inside" in g.files:
        assert np.array_equal(res.particle_inside_aabb, g["inside"].astype(bool))
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.vertices, res.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"], cmp
    assert cmp["max_rel_diff"] <= 1e-6, cmp  # only shared-face vertices may differ (by an ulp): "first patch wins"


@pytest.mark.parametrize("name", DIGEST)
def test_oracle_matches_reference_digest(oracle, name):
    g = load_golden(name)
    pts = golden_input(g)
    res = run_oracle(oracle, pts, golden_params(g))
    assert np.array_equal(res.particle_densities.view(np.uint32), g["densities"].view(np.uint32))
    ids, vs, tc = MC.canonicalize_geometric(res.vertices, res.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert ids.size == int(g["n_vertices"]) and tc.shape[0] == int(g["n_triangles"])
    assert hashlib.sha256(ids.astype(np.int64).tobytes()).hexdigest() == str(g["ids_sha256"])
    assert hashlib.sha256(tc.astype(np.int64).tobytes()).hexdigest() == str(g["triangles_sha256"])
    sel = g["sample_index"]
    assert np.array_equal(ids[sel], g["sample_ids"])
    d = np.abs(vs[sel].astype(np.float64) - g["sample_vertices"].astype(np.float64))
    assert d.max() <= 1e-6 * max(1.0, np.abs(g["sample_vertices"]).max())


def test_kat_known_answer(oracle):
    """tests/integration_tests/test_simple.rs:71-126: one particle => 6 vertices / 8 triangles, closed manifold."""
    g = load_golden("kat1")
    res = run_oracle(oracle, golden_input(g), golden_params(g))
    assert res.vertices.shape == (6, 3) and res.triangles.shape == (8, 3)
    assert MC.mesh_is_closed_manifold(res.triangles)
    assert abs(float(res.particle_densities[0]) - 20371.834) < 1e-2


def test_cubic_kernel_properties(oracle):
    """kernel.rs:143-180: compact support and unit integral (20^3 midpoint rule)."""
    h = 0.3
    assert oracle.kernel_evaluate(h, h) == 0.0
    assert oracle.kernel_evaluate(h, 2 * h) == 0.0
    assert oracle.kernel_evaluate(h, 0.999 * h) > 0.0
    n = 20
    dr = 2 * h / n
    c = (np.arange(n) + 0.5) * dr - h
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    r = np.sqrt(X ** 2 + Y ** 2 + Z ** 2).ravel()
    w = np.array([oracle.kernel_evaluate(h, ri) for ri in r], dtype=np.float64)
    assert abs(w.sum() * dr ** 3 - 1.0) < 1e-3


def test_mc_table_known_cases(oracle):
    """marching_cubes_lut.rs:375-451: case 'only corner 0 inside' => [3, 8, 0]; a case and its complement cut the same edges."""
    t = oracle.mc_table()
    assert t.shape == (256, 16)
    assert list(t[1][:4]) == [3, 8, 0, -1]
    assert t[0][0] == -1 and t[255][0] == -1
    ntri = (t >= 0).sum(axis=1) // 3
    assert ntri.max() == 5 and ntri.sum() == 820
    for c in range(256):
        assert set(t[c][t[c] >= 0]) == set(t[255 - c][t[255 - c] >= 0])


def test_oracle_mesh_closed_on_data(oracle):
    """test_full.rs:144-157: triangle count + closed/manifold for the double dam break."""
    g = load_golden("config1_double_dam_break")
    res = run_oracle(oracle, golden_input(g), golden_params(g))
    assert res.triangles.shape[0] == 66220 and res.vertices.shape[0] == 33026
    assert MC.mesh_is_closed_manifold(res.triangles)


def test_oracle_s1m_counts(oracle):
    """config 2 (1M uniform random, 8x over-dense): counts + densities of the reference."""
    g = load_golden("config2_s1m")
    res = run_oracle(oracle, golden_input(g), golden_params(g))
    assert res.vertices.shape[0] == int(g["n_vertices"]) == 76476
    assert res.triangles.shape[0] == int(g["n_triangles"]) == 152948
    sel = g["density_sample_index"]
    assert np.array_equal(res.particle_densities[sel].view(np.uint32), g["density_sample"].view(np.uint32))
    assert hashlib.sha256(res.particle_densities.tobytes()).hexdigest() == str(g["density_sha256"])


@pytest.mark.parametrize("name", ["neighbors_cube_2366_n16", "neighbors_config1"])
def test_oracle_neighbor_lists_match_reference(oracle, name):
    """global_neighborhood_list=True: per-particle neighbour lists incl. their ORDER equal the reference's
    (dense_subdomains.rs:617-639)."""
    g = load_golden(name)
    prm = golden_params(g)
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"],
                                      subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], global_neighborhood_list=True)
    res = oracle.reconstruct_surface(golden_input(g), par)
    assert np.array_equal(res.neighbor_ptr.astype(np.int64), g["row_ptr"])
    assert np.array_equal(res.neighbors.astype(np.int64), g["neighbors"].astype(np.int64))
    # symmetry of the relation d^2 < h^2 (test_neighborhood_search.rs checks the lists against a naive O(N^2) search)
    ptr, idx = g["row_ptr"], g["neighbors"].astype(np.int64)
    src = np.repeat(np.arange(ptr.size - 1), np.diff(ptr))
    a = set(zip(src.tolist(), idx.tolist()))
    assert all((j, i) in a for (i, j) in list(a)[:5000])


F64 = ["f64_kat1", "f64_cube_2366_n16", "f64_free_particles_125", "f64_config1", "f64_tank_small"]


@pytest.mark.parametrize("name", F64)
def test_oracle_f64_matches_reference(oracle, name):
    """reconstruct_surface::<i64, f64>: densities bit-identical (64-bit patterns), mesh identical."""
    g = load_golden(name)
    prm = golden_params(g)
    pts = golden_input(g).astype(np.float64)
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"],
                                      iso_surface_threshold=prm["iso_surface_threshold"],
                                      subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], dtype=np.float64)
    res = oracle.reconstruct_surface(pts, par)
    assert res.vertices.dtype == np.float64
    assert np.array_equal(res.particle_densities.view(np.uint64), g["densities"].view(np.uint64))
    assert np.array_equal(res.grid["aabb_min"].view(np.uint64), g["grid_min"].view(np.uint64))
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.vertices, res.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] <= 1e-14, cmp


GLOBAL = ["global_kat1", "global_edge_empty", "global_cube_8", "global_cube_2366", "global_cube_2366_auto_disable",
          "global_cube_2366_aabb", "global_free_particles_125", "global_config1", "global_f64_cube_2366", "global_f64_config1"]


def run_oracle_global(O, g):
    """Oracle run of a `global_*` golden (rows A14/A15): parameters as stored, dtype from the golden's arrays."""
    prm = golden_params(g)
    dt = g["densities"].dtype.type
    pts = golden_input(g).astype(dt)
    kw = {}
    if "aabb_min" in prm:
        kw = dict(aabb_min=np.asarray(prm["aabb_min"], dt), aabb_max=np.asarray(prm["aabb_max"], dt))
    par = O.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"],
                                 iso_surface_threshold=prm["iso_surface_threshold"], dtype=dt,
                                 subdomain_grid=prm.get("subdomain_grid", True),
                                 subdomain_grid_auto_disable=prm.get("subdomain_grid_auto_disable", False), **kw)
    return O.reconstruct_surface(pts, par), dt


@pytest.mark.parametrize("name", GLOBAL)
def test_oracle_global_strategy_matches_reference(oracle, name):
    """reconstruct_surface_global (reconstruction.rs:65-194, sequential functions): densities, neighbour lists and
    vertex coordinates bit-identical to the reference, same triangles; `subdomain_grid` is None."""
    g = load_golden(name)
    res, dt = run_oracle_global(oracle, g)
    U = np.uint32 if dt == np.float32 else np.uint64
    assert res.used_global_strategy and res.subdomain_grid is None
    assert np.array_equal(res.grid["n_cells"], g["n_cells"])
    assert np.array_equal(res.grid["aabb_min"].view(U), g["grid_min"].view(U))
    assert np.array_equal(res.grid["aabb_max"].view(U), g["grid_max"].view(U))
    assert np.array_equal(res.particle_densities.view(U), g["densities"].view(U))
    assert np.array_equal(res.neighbor_ptr.astype(np.int64), g["row_ptr"])
    assert np.array_equal(res.neighbors.astype(np.int64), g["neighbors"].astype(np.int64))
    if "inside" in g.files:
        assert np.array_equal(res.particle_inside_aabb, g["inside"].astype(bool))
    assert res.vertices.shape[0] == g["vertices"].shape[0] and res.triangles.shape[0] == g["triangles"].shape[0]
    if res.vertices.shape[0]:
        cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.vertices, res.triangles, g["grid_min"], g["cell_size"], g["n_points"])
        assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] == 0.0, cmp
        assert MC.mesh_is_closed_manifold(res.triangles)


def test_oracle_auto_disable_rule(oracle):
    """lib.rs:421-441: the decomposition is used iff max cells per dim > (1.2 n) as u32."""
    g = load_golden("global_cube_2366")
    pts = golden_input(g)
    for n_cubes, expect_global in ((64, True), (48, True), (41, False), (16, False)):
        # the domain has 50 cells along y: 50 > (1.2*41) as u32 = 49 -> decomposition; 50 > 57 false -> global
        par = oracle.make_params_relative(0.025, 2.0, 0.75, subdomain_num_cubes_per_dim=n_cubes, subdomain_grid_auto_disable=True)
        res = oracle.reconstruct_surface(pts, par)
        assert res.used_global_strategy == expect_global, n_cubes


def test_oracle_dense_marching_cubes_matches_reference(oracle):
    """pysplashsurf.marching_cubes on a dense array (rows A15): vertex coordinates bit-identical, same triangles; an
    array whose values equal the threshold hits the reference's triangulation error ("missing iso surface vertex")."""
    g = load_golden("marching_cubes_dense")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        vals = np.ascontiguousarray(g["values"].astype(dt))
        res = oracle.marching_cubes(vals, float(g["threshold"]), float(g["cube_size"]), g["translation"])
        assert np.array_equal(res.grid["aabb_min"], g["translation"].astype(dt))  # UniformGrid::new does not align the origin
        cmp = MC.compare_geometric(g["v_" + tag], g["t_" + tag].astype(np.int64), res.vertices, res.triangles, res.grid["aabb_min"], dt(g["cube_size"]),
                                   np.array(vals.shape))
        assert cmp["ids_equal"] and cmp["triangles_equal"] and cmp["max_rel_diff"] == 0.0, cmp
    with pytest.raises(RuntimeError, match="code 3"):
        oracle.marching_cubes(g["eq_values"], 1.0, 1.0)


def _single_cell_values():
    """marching_cubes.rs:352-361 (test_interpolate_cell_data): corner values of one unit cell, array[i][j][k]."""
    v = np.zeros((2, 2, 2), np.float64)
    for ijk, val in (([0, 0, 0], 0.0), ([1, 0, 0], 0.75), ([1, 1, 0], 1.0), ([0, 1, 0], 0.5), ([0, 0, 1], 0.0), ([1, 0, 1], 0.0), ([1, 1, 1], 1.0), ([0, 1, 1], 0.0)):
        v[tuple(ijk)] = val
    return v


# local edges 0, 3, 5, 6, 9, 11 of the cell (marching_cubes.rs:387-392) as global edge keys (lower point flat index * 3 + axis)
SINGLE_CELL_KEYS = [0, 1, 8, 9, 14, 16]


def test_oracle_single_cell_known_answer(oracle):
    """The reference's unit test of the narrow-band extraction (marching_cubes.rs:325-398): threshold 0.25 on one cell
    gives 6 iso-surface vertices on edges 0, 3, 5, 6, 9, 11."""
    res = oracle.marching_cubes(_single_cell_values(), 0.25, 1.0)
    assert res.vertices.shape[0] == 6 and sorted(int(k) for k in res.vertex_keys) == SINGLE_CELL_KEYS
    empty = oracle.marching_cubes(np.zeros((2, 2, 2)), 0.25, 1.0)
    assert empty.vertices.shape[0] == 0 and empty.triangles.shape[0] == 0


# ---- the reference's default arithmetic: Parameters::enable_simd = true (goldens: tools/gen_goldens_simd.py) ----
@pytest.mark.parametrize("name", ["simd_kat1", "simd_cube_2366_n16", "simd_config1_double_dam_break"])
@pytest.mark.parametrize("mode", [1, 2])
def test_oracle_simd_modes_match_reference_simd(oracle, name, mode):
    """Mode 1 (lane-by-lane restatement of density_grid_loop_avx, dense_subdomains.rs:991-1133, with its unfused remainder
    lanes and the scalar loop for sparse subdomains) reproduces the wheel's simd=True mesh up to the reference's own freedom
    on subdomain faces; mode 2 (that arithmetic applied uniformly = what the HIP library computes) keeps the topology and
    stays within the north-star tolerance."""
    g = load_golden(name)
    prm = golden_params(g)
    pts = golden_input(g)
    par = oracle.make_params_relative(prm["particle_radius"], prm["smoothing_length"], prm["cube_size"], iso_surface_threshold=prm["iso_surface_threshold"],
                                      subdomain_num_cubes_per_dim=prm["subdomain_num_cubes_per_dim"], simd=mode)
    res = oracle.reconstruct_surface(pts, par)
    assert hashlib.sha256(res.particle_densities.tobytes()).hexdigest() == str(g["density_sha256"])  # densities do not depend on simd
    cmp = MC.compare_geometric(g["vertices"], g["triangles"], res.vertices, res.triangles, g["grid_min"], g["cell_size"], g["n_points"])
    assert cmp["ids_equal"] and cmp["triangles_equal"], cmp
    assert cmp["max_rel_diff"] <= (1e-6 if mode == 1 else 1e-5), cmp
    if mode == 1:
        assert cmp["n_vertices_bit_equal"] >= 0.99 * g["vertices"].shape[0]


def test_oracle_avx_kernel_against_scalar_kernel(oracle):
    """kernel.rs:381-481 (the reference's AVX-vs-scalar kernel test): the (1 - q) form with sigma = 8/(pi h^3) equals the scalar
    cubic spline within 5e-6 absolute or 1e-5 relative, and vanishes outside the support."""
    import ctypes
    L = oracle.lib()
    L.so_avx_kernel_evaluate.argtypes = [ctypes.c_float, ctypes.c_float]
    L.so_avx_kernel_evaluate.restype = ctypes.c_float
    for h in (0.025, 0.1, 1.0, 3.0):
        for r in np.linspace(0.0, 1.25 * h, 101):
            a = float(L.so_avx_kernel_evaluate(np.float32(h), np.float32(r)))
            b = float(oracle.kernel_evaluate(h, r))
            assert abs(a - b) <= max(5e-6, 1e-5 * abs(b)), (h, r, a, b)
            if r > h:
                assert a == 0.0


def test_splat_lower_bound_polynomial_bounds_the_spline():
    """The splat's classification pass certifies 'inside' with u^3 (c0 + c1 u^2), u = max(1 - q^2, 0), as a lower bound of the cubic
    spline W(q) / sigma (kernel.rs:71-81 in the v = 1 - q form: 1 - 6 q^2 + 6 q^3 below 1/2, 2 (1 - q)^3 above): the constants the
    kernel is compiled with must satisfy g <= W on [0, 1] with a relative gap (1e-4 by construction) far above the grid's
    Lipschitz error, and hold most of the kernel's mass."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "splashsurf_amd", "csrc", "ss_kernels.hip")).read()
    c0 = float(np.float32(float(re.search(r"#define SS_BOUND_C0 ([0-9.eE+-]+)f", src).group(1))))
    c1 = float(np.float32(float(re.search(r"#define SS_BOUND_C1 ([0-9.eE+-]+)f", src).group(1))))
    q = np.linspace(0.0, 1.0, 200001)
    u = 1.0 - q * q
    w = np.where(q < 0.5, 1.0 - 6.0 * q * q + 6.0 * q ** 3, 2.0 * (1.0 - q) ** 3)
    g = c0 * u ** 3 + c1 * u ** 5
    assert np.all(g >= 0.0) and np.all(g <= w)
    low = q <= 0.9
    assert np.min((w - g)[low] / g[low]) > 5.0e-5
    # the grid cannot hide a violation below 0.9: the smallest gap there is far above (Lipschitz constant of w - g) x (grid step)
    assert np.min((w - g)[low]) > 10.0 * np.abs(np.gradient(w - g, q)).max() * (q[1] - q[0])
    # above 0.9: u = (1 - q)(1 + q) <= 2 (1 - q) and u <= 0.19, so g <= 8 (1 - q)^3 (c0 + 0.0361 c1) < 2 (1 - q)^3 = W
    assert 8.0 * (c0 + 0.0361 * c1) < 2.0
    mass = np.trapezoid(g * q * q, q) / np.trapezoid(w * q * q, q)
    assert mass > 0.95


def test_splat_certificate_quartic_bounds_the_spline():
    """Round 6: the certificate on the matrix pipe uses C4 u^4 <= W(q) / sigma, u = max(1 - q^2, 0) (splat_cert_record).  For q >= 1/2 the ratio
    2 (1 - q)^3 / (1 - q^2)^4 = 2 / ((1 - q)(1 + q)^4) has its minimum 2 / (0.4 * 1.6^4) = 0.762939... at q = 3/5; the constants the kernel and the
    host are compiled with must lie below it, the inner piece must stay above, and the bound holds 90 % of the kernel's mass."""
    import os
    import re
    root = os.path.join(os.path.dirname(__file__), "..", "splashsurf_amd", "csrc")
    c4 = float(np.float32(float(re.search(r"#define SS_CERT_C4 ([0-9.eE+-]+)f", open(os.path.join(root, "ss_kernels.hip")).read()).group(1))))
    c4_host = float(re.search(r"P\.cert_vscale = \(R\)\(\(double\)([0-9.]+) \* \(double\)sig", open(os.path.join(root, "ss_api.hip")).read()).group(1))
    assert c4_host == pytest.approx(c4, abs=1e-7) and c4 < 2.0 / (0.4 * 1.6 ** 4)
    q = np.linspace(0.0, 1.0, 400001)
    u = 1.0 - q * q
    w = np.where(q < 0.5, 1.0 - 6.0 * q * q + 6.0 * q ** 3, 2.0 * (1.0 - q) ** 3)
    g = c4 * u ** 4
    assert np.all(g <= w)
    low = q <= 0.99
    assert np.min((w - g)[low] / g[low]) > 1.0e-6  # (tight at q = 0.6 by construction: the gap there is 0.762939 / 0.76293 - 1 = 1.2e-5)
    assert abs(np.gradient(w - g, q)[np.argmin(np.abs(q - 0.6))]) < 1e-3  # the gap's minimum at q = 0.6 is a tangent minimum, not a crossing the grid could straddle
    mass = np.trapezoid(g * q * q, q) / np.trapezoid(w * q * q, q)
    assert 0.90 < mass < 0.905


def test_splat_certificate_f16_operands_stay_below_the_bilinear_form():
    """splat_cert_record / the B operands of splat_accumulate_block_wave: every operand of the 32 x 32 x 8 tile is an f16, the products are exact and summed
    in f32.  With the slack eps = cert_e1 (|px| + |py| + |pz|) + cert_e0 taken off a = 1 - |p|^2 (make_device_params) the tile's value must not exceed
    s (1 - |x - p|^2) for any entry / point pair -- checked here by restating the record in numpy f32 / f16 and the exact value in f64, for the cube-size /
    support ratios of the sweep (coordinates relative to the block's centre in units of h)."""
    rng = np.random.default_rng(11)
    f16, f32 = np.float16, np.float32
    worst = -1.0
    for ratio in (1.0 / 30.0, 0.125, 0.2, 0.45, 1.0, 0.11, 2.0):
        xm = 3.5 * ratio * (1.0 + 1.0e-5) + 1.0e-6
        r = 2.0 ** -10 * (1.0 + 2.0 ** -10)
        e1, e0 = f32(r * 2.0 * xm * (1.0 + 1.0e-6)), f32((r * 3.0 * xm * xm + 3.0e-5) * (1.0 + 1.0e-6))
        n = 4000
        p = ((rng.random((n, 3)) * 2.0 - 1.0) * (3.5 * ratio + 0.66)).astype(f32)          # entries within the near radius of the block's box
        x = ((rng.integers(0, 8, size=(n, 3)).astype(np.float64) - 3.5) * ratio).astype(f32)  # the block's points
        x = (x.astype(np.float64) * (1.0 + (rng.random((n, 3)) - 0.5) * 1e-6)).astype(f32)    # (f32 noise of far-from-origin scenes)
        s = (0.55 + 0.3 * rng.random(n)).astype(f32)
        eps = (e1 * ((np.abs(p[:, 0]) + np.abs(p[:, 1])) + np.abs(p[:, 2])) + e0).astype(f32)
        a = (((f32(1.0) - eps) - p[:, 0] * p[:, 0]) - (p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2])).astype(f32)
        sa = (s * a).astype(f32)
        sa_hi = sa.astype(f16)
        sa_lo = (sa - sa_hi.astype(f32)).astype(f32).astype(f16)
        s2 = (s + s).astype(f32)
        P3 = (s2[:, None] * p).astype(f32).astype(f16)
        ms = (-s).astype(f16)
        xh = x.astype(f16)
        xxh = (x * x).astype(f32).astype(f16)
        d = (sa_hi.astype(np.float64) + sa_lo.astype(np.float64) + (P3.astype(np.float64) * xh.astype(np.float64)).sum(1) + (ms.astype(np.float64)[:, None] * xxh.astype(np.float64)).sum(1))
        truth = s.astype(np.float64) * (1.0 - ((x.astype(np.float64) - p.astype(np.float64)) ** 2).sum(1))
        # the f32 accumulation inside the instruction: eight products of magnitude < 10 -> below 1e-5, which cert_e0 holds (3e-5 s >= 1.6e-5)
        worst = max(worst, float(np.max(d + 1.0e-5 - truth)))
        assert np.all(d + 1.0e-5 <= truth), (ratio, float(np.max(d - truth)))
    assert worst < 0.0


