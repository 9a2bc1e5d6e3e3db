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


class SoParams(C.Structure):
    _fields_ = [
        ("particle_radius", C.c_float),
        ("rest_density", C.c_float),
        ("compact_support_radius", C.c_float),
        ("cube_size", C.c_float),
        ("iso_surface_threshold", C.c_float),
        ("has_particle_aabb", C.c_int32),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("subdomain_num_cubes_per_dim", C.c_int32),
        ("num_threads", C.c_int32),
        ("global_neighborhood_list", C.c_int32),
    ]


class SoGrid(C.Structure):
    _fields_ = [
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("cell_size", C.c_float),
        ("n_points", C.c_int64 * 3),
        ("n_cells", C.c_int64 * 3),
    ]


class SoResult(C.Structure):
    _fields_ = [
        ("grid", SoGrid),
        ("subdomain_grid", SoGrid),
        ("n_input", C.c_uint64),
        ("n_particles", C.c_uint64),
        ("particle_densities", C.POINTER(C.c_float)),
        ("particle_inside_aabb", C.POINTER(C.c_uint8)),
        ("neighbor_ptr", C.POINTER(C.c_uint64)),
        ("neighbors", C.POINTER(C.c_uint64)),
        ("n_vertices", C.c_uint64),
        ("vertices", C.POINTER(C.c_float)),
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
    ]


class SoShard(C.Structure):
    _fields_ = [("domain_min", C.c_float * 3), ("domain_max", C.c_float * 3), ("sub_lo", C.c_int64 * 3), ("sub_hi", C.c_int64 * 3)]


def build(force=False):
    """Compile the oracle with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("splash_oracle.c", "splash_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.so_reconstruct_surface.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SoParams), C.POINTER(SoResult)]
        L.so_reconstruct_surface.restype = C.c_int
        L.so_result_free.argtypes = [C.POINTER(SoResult)]
        L.so_result_free.restype = None
        L.so_grid_for_reconstruction.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SoParams), C.POINTER(SoGrid)]
        L.so_grid_for_reconstruction.restype = C.c_int
        L.so_debug_levelset_subdomain.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SoParams), C.c_int64, C.c_void_p]
        L.so_debug_levelset_subdomain.restype = C.c_int64
        L.so_cubic_kernel_evaluate.argtypes = [C.c_float, C.c_float]
        L.so_cubic_kernel_evaluate.restype = C.c_float
        L.so_mc_table.argtypes = []
        L.so_mc_table.restype = C.POINTER(C.c_int8)
        L.so_classify_particle.argtypes = [C.POINTER(SoGrid), C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int]
        L.so_classify_particle.restype = C.c_int
        L.so_grid_for_domain.argtypes = [C.POINTER(SoParams), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(SoGrid), C.POINTER(SoGrid),
                                         C.POINTER(C.c_float)]
        L.so_grid_for_domain.restype = C.c_int
        L.so_shard_densities.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SoParams), C.POINTER(SoShard), C.c_void_p]
        L.so_shard_densities.restype = C.c_int
        L.so_shard_reconstruct.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SoParams), C.POINTER(SoShard), C.c_void_p, C.POINTER(SoResult)]
        L.so_shard_reconstruct.restype = C.c_int
        L.so_debug_shard_levelset.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(SoParams), C.POINTER(SoShard), C.c_void_p, C.c_int64, C.c_void_p]
        L.so_debug_shard_levelset.restype = C.c_int64
        _lib = L
    return _lib


def make_params(particle_radius, compact_support_radius, cube_size, rest_density=1000.0,
                iso_surface_threshold=0.6, aabb_min=None, aabb_max=None,
                subdomain_num_cubes_per_dim=64, num_threads=0, global_neighborhood_list=False):
    """Absolute-unit parameters (lib.rs:197-210). All values are rounded to f32 here."""
    p = SoParams()
    p.particle_radius = np.float32(particle_radius)
    p.rest_density = np.float32(rest_density)
    p.compact_support_radius = np.float32(compact_support_radius)
    p.cube_size = np.float32(cube_size)
    p.iso_surface_threshold = np.float32(iso_surface_threshold)
    if aabb_min is not None and aabb_max is not None:
        p.has_particle_aabb = 1
        for d in range(3):
            p.aabb_min[d] = np.float32(aabb_min[d])
            p.aabb_max[d] = np.float32(aabb_max[d])
    else:
        p.has_particle_aabb = 0
    p.subdomain_num_cubes_per_dim = int(subdomain_num_cubes_per_dim)
    p.num_threads = int(num_threads)
    p.global_neighborhood_list = 1 if global_neighborhood_list else 0
    return p


def make_params_relative(particle_radius, smoothing_length, cube_size, **kw):
    """Radius-relative parameters exactly as the reference's Python binding forms them
    (pysplashsurf/src/reconstruction.rs:171-193): products in f64, then cast to f32."""
    r = float(particle_radius)
    return make_params(r, np.float32(2.0 * float(smoothing_length) * r), np.float32(float(cube_size) * r), **kw)


def _grid_dict(g):
    return dict(
        aabb_min=np.array(list(g.aabb_min), dtype=np.float32),
        aabb_max=np.array(list(g.aabb_max), dtype=np.float32),
        cell_size=np.float32(g.cell_size),
        n_points=np.array(list(g.n_points), dtype=np.int64),
        n_cells=np.array(list(g.n_cells), dtype=np.int64),
    )


class OracleResult:
    pass


def _make_shard(domain_min, domain_max, sub_lo, sub_hi):
    s = SoShard()
    for d in range(3):
        s.domain_min[d] = np.float32(domain_min[d])
        s.domain_max[d] = np.float32(domain_max[d])
        s.sub_lo[d] = int(sub_lo[d])
        s.sub_hi[d] = int(sub_hi[d])
    return s


def grid_for_domain(params, domain_min, domain_max):
    g, sg, m = SoGrid(), SoGrid(), C.c_float()
    a = (C.c_float * 3)(*[float(np.float32(x)) for x in domain_min])
    b = (C.c_float * 3)(*[float(np.float32(x)) for x in domain_max])
    rc = lib().so_grid_for_domain(C.byref(params), a, b, C.byref(g), C.byref(sg), C.byref(m))
    if rc != 0:
        raise RuntimeError("so_grid_for_domain failed")
    return _grid_dict(g), _grid_dict(sg), float(m.value)


def shard_densities(xyz, params, domain_min, domain_max, sub_lo, sub_hi):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    rho = np.zeros(xyz.shape[0], dtype=np.float32)
    s = _make_shard(domain_min, domain_max, sub_lo, sub_hi)
    rc = lib().so_shard_densities(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(s), rho.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("so_shard_densities failed with code %d" % rc)
    return rho


def shard_reconstruct(xyz, rho, params, domain_min, domain_max, sub_lo, sub_hi):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    rho = np.ascontiguousarray(rho, dtype=np.float32)
    s = _make_shard(domain_min, domain_max, sub_lo, sub_hi)
    res = SoResult()
    rc = lib().so_shard_reconstruct(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(s), rho.ctypes.data_as(C.c_void_p),
                                    C.byref(res))
    if rc != 0:
        raise RuntimeError("so_shard_reconstruct failed with code %d" % rc)
    return _unpack(res)


def shard_levelset(xyz, rho, params, domain_min, domain_max, sub_lo, sub_hi, flat_subdomain):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    rho = np.ascontiguousarray(rho, dtype=np.float32)
    s = _make_shard(domain_min, domain_max, sub_lo, sub_hi)
    n = params.subdomain_num_cubes_per_dim + 1
    out = np.zeros((n, n, n), dtype=np.float32)
    cnt = lib().so_debug_shard_levelset(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(s),
                                        rho.ctypes.data_as(C.c_void_p), int(flat_subdomain), out.ctypes.data_as(C.c_void_p))
    return cnt, out


def reconstruct_surface(xyz, params):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    res = SoResult()
    rc = lib().so_reconstruct_surface(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(res))
    if rc != 0:
        raise RuntimeError("oracle so_reconstruct_surface failed with code %d" % rc)
    return _unpack(res)


def _unpack(res):
    out = OracleResult()
    try:
        nv, nt, n = int(res.n_vertices), int(res.n_triangles), int(res.n_particles)
        out.vertices = np.ctypeslib.as_array(res.vertices, shape=(nv * 3,)).copy().reshape(nv, 3) if nv else np.zeros((0, 3), np.float32)
        out.vertex_keys = np.ctypeslib.as_array(res.vertex_keys, shape=(nv,)).copy() if nv else np.zeros((0,), np.uint64)
        out.triangles = np.ctypeslib.as_array(res.triangles, shape=(nt * 3,)).copy().reshape(nt, 3) if nt else np.zeros((0, 3), np.uint64)
        out.particle_densities = np.ctypeslib.as_array(res.particle_densities, shape=(n,)).copy() if n else np.zeros((0,), np.float32)
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
        out.grid = _grid_dict(res.grid)
        out.subdomain_grid = _grid_dict(res.subdomain_grid)
        out.n_subdomains = int(res.n_subdomains)
        out.n_subdomain_particles = int(res.n_subdomain_particles)
        out.timings = dict(total=res.t_total, decomposition=res.t_decomposition, density=res.t_density,
                           reconstruction=res.t_reconstruction, stitching=res.t_stitching)
        out.threads_used = int(res.threads_used)
    finally:
        lib().so_result_free(C.byref(res))
    return out


def grid_for_reconstruction(xyz, params):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    g = SoGrid()
    rc = lib().so_grid_for_reconstruction(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params), C.byref(g))
    if rc != 0:
        raise RuntimeError("oracle grid construction failed")
    return _grid_dict(g)


def levelset_subdomain(xyz, params, flat_subdomain):
    xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
    n = params.subdomain_num_cubes_per_dim + 1
    out = np.zeros((n, n, n), dtype=np.float32)
    cnt = lib().so_debug_levelset_subdomain(xyz.ctypes.data_as(C.c_void_p), xyz.shape[0], C.byref(params),
                                            int(flat_subdomain), out.ctypes.data_as(C.c_void_p))
    return cnt, out


def kernel_evaluate(h, r):
    return np.float32(lib().so_cubic_kernel_evaluate(np.float32(h), np.float32(r)))


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
