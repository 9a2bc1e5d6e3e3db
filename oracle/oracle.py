"""ctypes binding of the CPU oracle (oracle/splash_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package `splashsurf_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsplash_oracle.so")


def _make_structs(real):
    class Params(C.Structure):
        _fields_ = [
            ("particle_radius", real),
            ("rest_density", real),
            ("compact_support_radius", real),
            ("cube_size", real),
            ("iso_surface_threshold", real),
            ("has_particle_aabb", C.c_int32),
            ("aabb_min", real * 3),
            ("aabb_max", real * 3),
            ("subdomain_num_cubes_per_dim", C.c_int32),
            ("num_threads", C.c_int32),
            ("global_neighborhood_list", C.c_int32),
            ("global_strategy", C.c_int32),
            ("enable_simd", C.c_int32),
        ]

    class Grid(C.Structure):
        _fields_ = [
            ("aabb_min", real * 3),
            ("aabb_max", real * 3),
            ("cell_size", real),
            ("n_points", C.c_int64 * 3),
            ("n_cells", C.c_int64 * 3),
        ]

    class Result(C.Structure):
        _fields_ = [
            ("grid", Grid),
            ("subdomain_grid", Grid),
            ("n_input", C.c_uint64),
            ("n_particles", C.c_uint64),
            ("particle_densities", C.POINTER(real)),
            ("particle_inside_aabb", C.POINTER(C.c_uint8)),
            ("neighbor_ptr", C.POINTER(C.c_uint64)),
            ("neighbors", C.POINTER(C.c_uint64)),
            ("n_vertices", C.c_uint64),
            ("vertices", C.POINTER(real)),
            ("vertex_keys", C.POINTER(C.c_uint64)),
            ("n_triangles", C.c_uint64),
            ("triangles", C.POINTER(C.c_uint64)),
            ("n_subdomains", C.c_int64),
            ("n_subdomain_particles", C.c_uint64),
            ("t_total", C.c_double),
            ("t_decomposition", C.c_double),
            ("t_density", C.c_double),
            ("t_reconstruction", C.c_double),
            ("t_stitching", C.c_double),
            ("threads_used", C.c_int32),
            ("used_global_strategy", C.c_int32),
            ("global_levelset", C.POINTER(real)),
        ]

    class Shard(C.Structure):
        _fields_ = [("domain_min", real * 3), ("domain_max", real * 3), ("sub_lo", C.c_int64 * 3), ("sub_hi", C.c_int64 * 3)]

    return Params, Grid, Result, Shard


SoParams, SoGrid, SoResult, SoShard = _make_structs(C.c_float)          # f32 instantiation (so_*)
SoParams64, SoGrid64, SoResult64, SoShard64 = _make_structs(C.c_double)  # f64 instantiation (so64_*)


def _flavour(obj):
    """(prefix, numpy dtype, ctypes real, struct classes) for a params object or a dtype."""
    if isinstance(obj, SoParams64) or obj is np.float64 or obj == np.dtype(np.float64):
        return "so64_", np.float64, C.c_double, (SoParams64, SoGrid64, SoResult64, SoShard64)
    return "so_", np.float32, C.c_float, (SoParams, SoGrid, SoResult, SoShard)


def build(force=False):
    """Compile the oracle with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("splash_oracle.c", "splash_post.c", "splash_oracle.h", "splash_oracle_decl.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        for pre, (Pm, Gr, Rs, Sh), real in (("so_", (SoParams, SoGrid, SoResult, SoShard), C.c_float),
                                             ("so64_", (SoParams64, SoGrid64, SoResult64, SoShard64), C.c_double)):
            f = lambda n: getattr(L, pre + n)
            f("reconstruct_surface").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Pm), C.POINTER(Rs)]
            f("reconstruct_surface").restype = C.c_int
            f("result_free").argtypes = [C.POINTER(Rs)]
            f("result_free").restype = None
            f("grid_for_reconstruction").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Pm), C.POINTER(Gr)]
            f("grid_for_reconstruction").restype = C.c_int
            f("debug_levelset_subdomain").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Pm), C.c_int64, C.c_void_p]
            f("debug_levelset_subdomain").restype = C.c_int64
            f("cubic_kernel_evaluate").argtypes = [real, real]
            f("cubic_kernel_evaluate").restype = real
            f("mc_table").argtypes = []
            f("mc_table").restype = C.POINTER(C.c_int8)
            f("classify_particle").argtypes = [C.POINTER(Gr), real, C.POINTER(real), C.POINTER(C.c_int64), C.c_int]
            f("classify_particle").restype = C.c_int
            f("grid_for_domain").argtypes = [C.POINTER(Pm), C.POINTER(real), C.POINTER(real), C.POINTER(Gr), C.POINTER(Gr), C.POINTER(real)]
            f("grid_for_domain").restype = C.c_int
            f("shard_densities").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Pm), C.POINTER(Sh), C.c_void_p]
            f("shard_densities").restype = C.c_int
            f("shard_reconstruct").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Pm), C.POINTER(Sh), C.c_void_p, C.POINTER(Rs)]
            f("shard_reconstruct").restype = C.c_int
            f("debug_shard_levelset").argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Pm), C.POINTER(Sh), C.c_void_p, C.c_int64, C.c_void_p]
            f("debug_shard_levelset").restype = C.c_int64
        _lib = L
    return _lib


def make_params(particle_radius, compact_support_radius, cube_size, rest_density=1000.0,
                iso_surface_threshold=0.6, aabb_min=None, aabb_max=None,
                subdomain_num_cubes_per_dim=64, num_threads=0, global_neighborhood_list=False, dtype=np.float32,
                subdomain_grid=True, subdomain_grid_auto_disable=False, simd=0):
    """Absolute-unit parameters (lib.rs:197-210), rounded to `dtype` (float32: so_*, float64: so64_*).
    subdomain_grid=False: SpatialDecomposition::None; subdomain_grid_auto_disable=True: the reference's
    default rule (global strategy when the domain has <= 1.2 n cells per dimension, lib.rs:421-441).
    simd: 0 = scalar level-set loop (Parameters::enable_simd = false); 1 = the reference's AVX2+FMA loop for dense
    subdomains, f32 only (enable_simd = true on x86-64); 2 = that arithmetic applied uniformly (splash_oracle_decl.h)."""
    _, npdt, _, (Pm, _, _, _) = _flavour(np.dtype(dtype))
    p = Pm()
    p.particle_radius = npdt(particle_radius)
    p.rest_density = npdt(rest_density)
    p.compact_support_radius = npdt(compact_support_radius)
    p.cube_size = npdt(cube_size)
    p.iso_surface_threshold = npdt(iso_surface_threshold)
    if aabb_min is not None and aabb_max is not None:
        p.has_particle_aabb = 1
        for d in range(3):
            p.aabb_min[d] = npdt(aabb_min[d])
            p.aabb_max[d] = npdt(aabb_max[d])
    else:
        p.has_particle_aabb = 0
    p.subdomain_num_cubes_per_dim = int(subdomain_num_cubes_per_dim)
    p.num_threads = int(num_threads)
    p.global_neighborhood_list = 1 if global_neighborhood_list else 0
    p.global_strategy = 1 if not subdomain_grid else (2 if subdomain_grid_auto_disable else 0)
    p.enable_simd = int(simd)
    return p


def make_params_relative(particle_radius, smoothing_length, cube_size, dtype=np.float32, **kw):
    """Radius-relative parameters exactly as the reference's Python binding forms them
    (pysplashsurf/src/reconstruction.rs:171-193): products in f64, then converted to the Real type."""
    r = float(particle_radius)
    npdt = np.dtype(dtype).type
    return make_params(r, npdt(2.0 * float(smoothing_length) * r), npdt(float(cube_size) * r), dtype=dtype, **kw)


def _grid_dict(g, npdt=np.float32):
    return dict(
        aabb_min=np.array(list(g.aabb_min), dtype=npdt),
        aabb_max=np.array(list(g.aabb_max), dtype=npdt),
        cell_size=npdt(g.cell_size),
        n_points=np.array(list(g.n_points), dtype=np.int64),
        n_cells=np.array(list(g.n_cells), dtype=np.int64),
    )


class OracleResult:
    pass


def _fn(params, name):
    return getattr(lib(), _flavour(params)[0] + name)


def _xyz(xyz, params):
    return np.ascontiguousarray(xyz, dtype=_flavour(params)[1]).reshape(-1, 3)


def _make_shard(params, domain_min, domain_max, sub_lo, sub_hi):
    _, npdt, _, (_, _, _, Sh) = _flavour(params)
    s = Sh()
    for d in range(3):
        s.domain_min[d] = npdt(domain_min[d])
        s.domain_max[d] = npdt(domain_max[d])
        s.sub_lo[d] = int(sub_lo[d])
        s.sub_hi[d] = int(sub_hi[d])
    return s


def grid_for_domain(params, domain_min, domain_max):
    _, npdt, creal, (_, Gr, _, _) = _flavour(params)
    g, sg, m = Gr(), Gr(), creal()
    a = (creal * 3)(*[float(npdt(x)) for x in domain_min])
    b = (creal * 3)(*[float(npdt(x)) for x in domain_max])
    rc = _fn(params, "grid_for_domain")(C.byref(params), a, b, C.byref(g), C.byref(sg), C.byref(m))
    if rc != 0:
        raise RuntimeError("so_grid_for_domain failed")
    return _grid_dict(g, npdt), _grid_dict(sg, npdt), float(m.value)


def shard_densities(xyz, params, domain_min, domain_max, sub_lo, sub_hi):
    xyz = _xyz(xyz, params)
    rho = np.zeros(xyz.shape[0], dtype=xyz.dtype)
    s = _make_shard(params, domain_min, domain_max, sub_lo, sub_hi)
    rc = _fn(params, "shard_densities")(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(s), rho.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("so_shard_densities failed with code %d" % rc)
    return rho


def shard_reconstruct(xyz, rho, params, domain_min, domain_max, sub_lo, sub_hi):
    xyz = _xyz(xyz, params)
    rho = np.ascontiguousarray(rho, dtype=xyz.dtype)
    s = _make_shard(params, domain_min, domain_max, sub_lo, sub_hi)
    res = _flavour(params)[3][2]()
    rc = _fn(params, "shard_reconstruct")(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(s),
                                          rho.ctypes.data_as(C.c_void_p), C.byref(res))
    if rc != 0:
        raise RuntimeError("so_shard_reconstruct failed with code %d" % rc)
    return _unpack(res, params)


def shard_levelset(xyz, rho, params, domain_min, domain_max, sub_lo, sub_hi, flat_subdomain):
    xyz = _xyz(xyz, params)
    rho = np.ascontiguousarray(rho, dtype=xyz.dtype)
    s = _make_shard(params, domain_min, domain_max, sub_lo, sub_hi)
    n = params.subdomain_num_cubes_per_dim + 1
    out = np.zeros((n, n, n), dtype=xyz.dtype)
    cnt = _fn(params, "debug_shard_levelset")(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(s),
                                              rho.ctypes.data_as(C.c_void_p), int(flat_subdomain), out.ctypes.data_as(C.c_void_p))
    return cnt, out


def reconstruct_surface(xyz, params):
    xyz = _xyz(xyz, params)
    res = _flavour(params)[3][2]()
    rc = _fn(params, "reconstruct_surface")(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(res))
    if rc != 0:
        raise RuntimeError("oracle so_reconstruct_surface failed with code %d" % rc)
    return _unpack(res, params)


def _unpack(res, params):
    npdt = _flavour(params)[1]
    out = OracleResult()
    try:
        nv, nt, n = int(res.n_vertices), int(res.n_triangles), int(res.n_particles)
        out.vertices = np.ctypeslib.as_array(res.vertices, shape=(nv * 3,)).copy().reshape(nv, 3) if nv else np.zeros((0, 3), npdt)
        out.vertex_keys = np.ctypeslib.as_array(res.vertex_keys, shape=(nv,)).copy() if nv else np.zeros((0,), np.uint64)
        out.triangles = np.ctypeslib.as_array(res.triangles, shape=(nt * 3,)).copy().reshape(nt, 3) if nt else np.zeros((0, 3), np.uint64)
        out.particle_densities = np.ctypeslib.as_array(res.particle_densities, shape=(n,)).copy() if n else np.zeros((0,), npdt)
        if res.neighbor_ptr:
            out.neighbor_ptr = np.ctypeslib.as_array(res.neighbor_ptr, shape=(n + 1,)).copy()
            m = int(out.neighbor_ptr[-1])
            out.neighbors = np.ctypeslib.as_array(res.neighbors, shape=(m,)).copy() if m else np.zeros(0, np.uint64)
        else:
            out.neighbor_ptr = out.neighbors = None
        if res.particle_inside_aabb:
            ni = int(res.n_input)
            out.particle_inside_aabb = np.ctypeslib.as_array(res.particle_inside_aabb, shape=(ni,)).copy().astype(bool) if ni else np.zeros((0,), bool)
        else:
            out.particle_inside_aabb = None
        out.grid = _grid_dict(res.grid, npdt)
        out.subdomain_grid = _grid_dict(res.subdomain_grid, npdt)
        out.n_subdomains = int(res.n_subdomains)
        out.n_subdomain_particles = int(res.n_subdomain_particles)
        out.timings = dict(total=res.t_total, decomposition=res.t_decomposition, density=res.t_density,
                           reconstruction=res.t_reconstruction, stitching=res.t_stitching)
        out.threads_used = int(res.threads_used)
        out.used_global_strategy = bool(res.used_global_strategy)
        if out.used_global_strategy:
            out.subdomain_grid = None
            npts = [int(x) for x in res.grid.n_points]
            tot = npts[0] * npts[1] * npts[2]
            out.global_levelset = (np.ctypeslib.as_array(res.global_levelset, shape=(tot,)).copy().reshape(npts)
                                   if res.global_levelset and tot else None)
        else:
            out.global_levelset = None
    finally:
        _fn(params, "result_free")(C.byref(res))
    return out


def marching_cubes(values, iso_surface_threshold, cube_size, translation=None):
    """pysplashsurf.marching_cubes on a dense array; raises RuntimeError(code) on the reference's error paths."""
    vals = np.ascontiguousarray(values)
    assert vals.ndim == 3 and vals.dtype in (np.float32, np.float64)
    pre, npdt, creal, (_, _, Rs, _) = _flavour(vals.dtype)
    fn = getattr(lib(), pre + "marching_cubes")
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_int64), creal, creal, C.POINTER(creal), C.POINTER(Rs)]
    fn.restype = C.c_int
    res = Rs()
    npts = (C.c_int64 * 3)(*[int(x) for x in vals.shape])
    tr = (creal * 3)(*([float(npdt(x)) for x in translation] if translation is not None else [0.0, 0.0, 0.0]))
    rc = fn(vals.ctypes.data_as(C.c_void_p), npts, creal(float(npdt(iso_surface_threshold))), creal(float(npdt(cube_size))), tr, C.byref(res))
    if rc != 0:
        getattr(lib(), pre + "result_free").argtypes = [C.POINTER(Rs)]
        getattr(lib(), pre + "result_free")(C.byref(res))
        raise RuntimeError("oracle marching_cubes failed with code %d" % rc)
    out = OracleResult()
    try:
        nv, nt = int(res.n_vertices), int(res.n_triangles)
        out.vertices = np.ctypeslib.as_array(res.vertices, shape=(nv * 3,)).copy().reshape(nv, 3) if nv else np.zeros((0, 3), npdt)
        out.vertex_keys = np.ctypeslib.as_array(res.vertex_keys, shape=(nv,)).copy() if nv else np.zeros((0,), np.uint64)
        out.triangles = np.ctypeslib.as_array(res.triangles, shape=(nt * 3,)).copy().reshape(nt, 3) if nt else np.zeros((0, 3), np.uint64)
        out.grid = _grid_dict(res.grid, npdt)
    finally:
        fr = getattr(lib(), pre + "result_free")
        fr.argtypes = [C.POINTER(Rs)]
        fr(C.byref(res))
    return out


def grid_for_reconstruction(xyz, params):
    xyz = _xyz(xyz, params)
    npdt = _flavour(params)[1]
    g = _flavour(params)[3][1]()
    rc = _fn(params, "grid_for_reconstruction")(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(g))
    if rc != 0:
        raise RuntimeError("oracle grid construction failed")
    return _grid_dict(g, npdt)


def levelset_subdomain(xyz, params, flat_subdomain):
    xyz = _xyz(xyz, params)
    n = params.subdomain_num_cubes_per_dim + 1
    out = np.zeros((n, n, n), dtype=xyz.dtype)
    cnt = _fn(params, "debug_levelset_subdomain")(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params),
                                                  int(flat_subdomain), out.ctypes.data_as(C.c_void_p))
    return cnt, out


def kernel_evaluate(h, r, dtype=np.float32):
    pre, npdt, _, _ = _flavour(np.dtype(dtype))
    return npdt(getattr(lib(), pre + "cubic_kernel_evaluate")(npdt(h), npdt(r)))


def mc_table():
    return np.ctypeslib.as_array(lib().so_mc_table(), shape=(256, 16)).copy()


def canonical_mesh(vertices, vertex_keys, triangles):
    """Sort vertices by global edge key and triangles lexicographically (rotation-normalised)."""
    order = np.argsort(vertex_keys, kind="stable")
    inv = np.empty_like(order)
    inv[order] = np.arange(order.size)
    v = vertices[order]
    k = vertex_keys[order]
    t = inv[triangles.astype(np.int64)] if triangles.size else triangles.astype(np.int64).reshape(0, 3)
    if t.size:
        # rotate each triangle so that its smallest index comes first (keeps orientation)
        amin = np.argmin(t, axis=1)
        rows = np.arange(t.shape[0])
        t = np.stack([t[rows, amin], t[rows, (amin + 1) % 3], t[rows, (amin + 2) % 3]], axis=1)
        t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    return v, k, t


# ---------------------------------------------------------------------------------------------------
# post-processing oracle (oracle/splash_post.c)
# ---------------------------------------------------------------------------------------------------
def _post(name, dtype):
    L = lib()
    return getattr(L, ("so64_post_" if np.dtype(dtype) == np.float64 else "so_post_") + name)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def post_vertex_connectivity(n_vertices, triangles):
    """mesh.rs:290-306 -> (row_ptr uint64[V+1], neighbors uint32[M])"""
    tris = np.ascontiguousarray(triangles, dtype=np.uint64).reshape(-1, 3)
    row, nb = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint32)()
    fn = _post("vertex_connectivity", np.float32)
    fn.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint32))]
    fn.restype = C.c_int
    fn(int(n_vertices), _vp(tris), tris.shape[0], C.byref(row), C.byref(nb))
    free = _post("free", np.float32)
    free.argtypes = [C.c_void_p]
    free.restype = None
    row_np = np.ctypeslib.as_array(row, shape=(int(n_vertices) + 1,)).copy()
    m = int(row_np[-1])
    nb_np = np.ctypeslib.as_array(nb, shape=(max(m, 1),)).copy()[:m]
    free(C.cast(row, C.c_void_p))
    free(C.cast(nb, C.c_void_p))
    return row_np, nb_np


def post_vertex_normals(vertices, triangles):
    v = np.ascontiguousarray(vertices)
    assert v.dtype in (np.float32, np.float64)
    tris = np.ascontiguousarray(triangles, dtype=np.uint64).reshape(-1, 3)
    out = np.zeros_like(v)
    fn = _post("vertex_normals", v.dtype)
    fn.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    fn.restype = None
    fn(_vp(v), v.shape[0], _vp(tris), tris.shape[0], _vp(out))
    return out


def post_laplacian_smoothing(vertices, row_ptr, nbrs, iterations, beta, weights):
    v = np.ascontiguousarray(vertices).copy()
    creal = C.c_double if v.dtype == np.float64 else C.c_float
    row = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    nb = np.ascontiguousarray(nbrs, dtype=np.uint32)
    w = np.ascontiguousarray(weights, dtype=v.dtype)
    fn = _post("laplacian_smoothing", v.dtype)
    fn.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, creal, C.c_void_p]
    fn.restype = None
    fn(_vp(v), v.shape[0], _vp(row), _vp(nb), int(iterations), creal(float(v.dtype.type(beta))), _vp(w))
    return v


def post_smooth_normals(normals, row_ptr, nbrs, iterations):
    nrm = np.ascontiguousarray(normals).copy()
    row = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    nb = np.ascontiguousarray(nbrs, dtype=np.uint32)
    fn = _post("smooth_normals", nrm.dtype)
    fn.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
    fn.restype = None
    fn(_vp(nrm), nrm.shape[0], _vp(row), _vp(nb), int(iterations))
    return nrm


def post_weighted_neighbor_counts(xyz, nb_ptr, nb_idx, h):
    x = np.ascontiguousarray(xyz)
    creal = C.c_double if x.dtype == np.float64 else C.c_float
    ptr = np.ascontiguousarray(nb_ptr, dtype=np.uint64)
    idx = np.ascontiguousarray(nb_idx, dtype=np.uint64)
    out = np.zeros(x.shape[0], dtype=x.dtype)
    fn = _post("weighted_neighbor_counts", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, creal, C.c_void_p]
    fn.restype = None
    fn(_vp(x), x.shape[0], _vp(ptr), _vp(idx), creal(float(x.dtype.type(h))), _vp(out))
    return out


def post_smoothing_weights(wnn, normalization):
    a = np.ascontiguousarray(wnn)
    creal = C.c_double if a.dtype == np.float64 else C.c_float
    out = np.zeros_like(a)
    fn = _post("smoothing_weights", a.dtype)
    fn.argtypes = [C.c_void_p, C.c_uint64, creal, C.c_void_p]
    fn.restype = None
    fn(_vp(a), a.shape[0], creal(float(a.dtype.type(normalization))), _vp(out))
    return out


def post_sph_interpolate(xyz, rho, rest_mass, h, values, points, first_order_correction):
    x = np.ascontiguousarray(xyz)
    creal = C.c_double if x.dtype == np.float64 else C.c_float
    r = np.ascontiguousarray(rho, dtype=x.dtype)
    vals = np.ascontiguousarray(values, dtype=x.dtype)
    dim = 1 if vals.ndim == 1 else int(vals.shape[1])
    pts = np.ascontiguousarray(points, dtype=x.dtype)
    out = np.zeros((pts.shape[0],) if dim == 1 else (pts.shape[0], dim), dtype=x.dtype)
    fn = _post("sph_interpolate", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, creal, creal, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    fn.restype = None
    fn(_vp(x), _vp(r), x.shape[0], creal(float(x.dtype.type(rest_mass))), creal(float(x.dtype.type(h))), _vp(vals), dim, _vp(pts), pts.shape[0],
       1 if first_order_correction else 0, _vp(out))
    return out


def post_sph_normals(xyz, rho, rest_mass, h, points):
    x = np.ascontiguousarray(xyz)
    creal = C.c_double if x.dtype == np.float64 else C.c_float
    r = np.ascontiguousarray(rho, dtype=x.dtype)
    pts = np.ascontiguousarray(points, dtype=x.dtype)
    out = np.zeros_like(pts)
    fn = _post("sph_normals", x.dtype)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, creal, creal, C.c_void_p, C.c_uint64, C.c_void_p]
    fn.restype = None
    fn(_vp(x), _vp(r), x.shape[0], creal(float(x.dtype.type(rest_mass))), creal(float(x.dtype.type(h))), _vp(pts), pts.shape[0], _vp(out))
    return out
