"""Python host of the MI355X surface reconstruction: ctypes over the C ABI (include/splashsurf_hip.h).

Mirrors the reference's Python entry point `pysplashsurf.reconstruct_surface`
(pysplashsurf/src/reconstruction.rs:135-207): same keyword names, same units (smoothing length and
cube size RELATIVE to the particle radius, products formed in f64 and cast to f32,
reconstruction.rs:171-193), same result attributes (`.mesh.vertices`, `.mesh.triangles` as uint64,
`.grid`, `.particle_densities`, `.particle_inside_aabb`).

There is NO CPU fallback: if libsplashsurf_hip.so is missing or no GPU is visible the call fails loudly.
"""
import ctypes as C
import weakref
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libsplashsurf_hip.so"


class SplashsurfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("splashsurf_hip error %d: %s" % (status, message))
        self.status = status


class GridConstructionError(SplashsurfError):
    """ReconstructionError::GridConstructionError (lib.rs:291-296)"""


class _Params(C.Structure):
    _fields_ = [
        ("particle_radius", C.c_float),
        ("rest_density", C.c_float),
        ("compact_support_radius", C.c_float),
        ("cube_size", C.c_float),
        ("iso_surface_threshold", C.c_float),
        ("has_particle_aabb", C.c_int32),
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("enable_multi_threading", C.c_int32),
        ("enable_simd", C.c_int32),
        ("decomposition", C.c_int32),
        ("subdomain_num_cubes_per_dim", C.c_uint32),
        ("auto_disable", C.c_int32),
        ("global_neighborhood_list", C.c_int32),
    ]


class _Grid(C.Structure):
    _fields_ = [
        ("aabb_min", C.c_float * 3),
        ("aabb_max", C.c_float * 3),
        ("cell_size", C.c_float),
        ("n_points", C.c_int64 * 3),
        ("n_cells", C.c_int64 * 3),
    ]


class _Params64(C.Structure):
    _fields_ = [
        ("particle_radius", C.c_double),
        ("rest_density", C.c_double),
        ("compact_support_radius", C.c_double),
        ("cube_size", C.c_double),
        ("iso_surface_threshold", C.c_double),
        ("has_particle_aabb", C.c_int32),
        ("aabb_min", C.c_double * 3),
        ("aabb_max", C.c_double * 3),
        ("enable_multi_threading", C.c_int32),
        ("enable_simd", C.c_int32),
        ("decomposition", C.c_int32),
        ("subdomain_num_cubes_per_dim", C.c_uint32),
        ("auto_disable", C.c_int32),
        ("global_neighborhood_list", C.c_int32),
    ]


class _Grid64(C.Structure):
    _fields_ = [
        ("aabb_min", C.c_double * 3),
        ("aabb_max", C.c_double * 3),
        ("cell_size", C.c_double),
        ("n_points", C.c_int64 * 3),
        ("n_cells", C.c_int64 * 3),
    ]


class _Stats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double),
        ("ms_upload", C.c_double),
        ("ms_aabb_grid", C.c_double),
        ("ms_decomposition", C.c_double),
        ("ms_density", C.c_double),
        ("ms_levelset", C.c_double),
        ("ms_levelset_prepare", C.c_double),
        ("ms_marching_cubes", C.c_double),
        ("ms_stitching", C.c_double),
        ("n_particles", C.c_uint64),
        ("n_vertices", C.c_uint64),
        ("n_triangles", C.c_uint64),
        ("n_active_blocks", C.c_uint64),
        ("n_block_candidates", C.c_uint64),
        ("fast_div_verified", C.c_uint64),
        ("levelset_kernel_launches", C.c_uint64),
        ("bytes_device_peak", C.c_uint64),
        ("ms_levelset_gather", C.c_double),
        ("ms_levelset_accumulate", C.c_double),
        ("n_large_tile_blocks", C.c_uint64),
        ("arith_mode", C.c_uint64),
        ("bytes_tile_arena", C.c_uint64),
        ("bytes_tile_arena_reserved", C.c_uint64),
        ("n_certified_subblocks", C.c_uint64),
        ("n_truncated_blocks", C.c_uint64),
        ("n_completed_blocks", C.c_uint64),
        ("ms_levelset_accumulate_pass2", C.c_double),
        ("n_mc_blocks", C.c_uint64),
        ("ms_density_kernel", C.c_double),
        ("ms_mc_count", C.c_double),
        ("ms_mc_emit", C.c_double),
        ("n_host_waits", C.c_uint64),
    ]


_lib = None


def library_path():
    # SPLASHSURF_HIP_LIB: another build of the same library (kernel variants compared side by side by tools/ab_kernels.sh)
    return os.environ.get("SPLASHSURF_HIP_LIB") or os.path.join(_HERE, _LIB_NAME)


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64
    (same SONAMEs as /opt/rocm's).  If torch is installed, load ITS copies first (RTLD_GLOBAL) so that
    this library and torch (device tensors, streams, RCCL) share a single runtime regardless of import
    order; without torch the system ROCm runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("librocprofiler-register.so", "libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load_library():
    """Load libsplashsurf_hip.so (built in-tree by __graft_entry__.build()). Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback." % path)
    L = C.CDLL(path)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int32
    P = C.POINTER
    L.ss_abi_version.restype = C.c_int
    L.ss_context_create.argtypes = [C.c_int, P(vp)]
    L.ss_context_destroy.argtypes = [vp]
    L.ss_context_destroy.restype = None
    L.ss_last_error.argtypes = [vp]
    L.ss_last_error.restype = C.c_char_p
    L.ss_last_error_detail.argtypes = [vp]
    L.ss_context_set_stream.argtypes = [vp, vp]
    L.ss_reconstruct_surface_f32.argtypes = [vp, vp, u64, P(_Params), P(vp)]
    L.ss_reconstruct_surface_inplace_f32.argtypes = [vp, vp, u64, P(_Params), vp]
    L.ss_grid_for_reconstruction_f32.argtypes = [vp, vp, u64, P(_Params), P(_Grid)]
    L.ss_result_create.argtypes = [vp, P(vp)]
    L.ss_result_free.argtypes = [vp]
    L.ss_result_free.restype = None
    L.ss_result_counts.argtypes = [vp, P(u64), P(u64)]
    L.ss_result_vertices.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_triangles.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_triangles_u32.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_grid.argtypes = [vp, P(_Grid)]
    L.ss_result_subdomain_grid.argtypes = [vp, P(_Grid), P(i32)]
    L.ss_result_particle_densities.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_particle_inside_aabb.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_stats.argtypes = [vp, P(_Stats)]
    L.ss_result_particle_neighbors.argtypes = [vp, P(vp), P(vp), P(u64)]
    L.ss_result_device_vertices.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_device_triangles_u32.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_device_particle_densities.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_vertex_keys.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_levelset_box.argtypes = [vp, P(C.c_int64), P(C.c_int64), vp]
    L.ss_result_subdomain_stats.argtypes = [vp, P(u64), P(u64)]
    L.ss_reconstruct_surface_f64.argtypes = [vp, vp, u64, P(_Params64), P(vp)]
    L.ss_reconstruct_surface_inplace_f64.argtypes = [vp, vp, u64, P(_Params64), vp]
    L.ss_grid_for_reconstruction_f64.argtypes = [vp, vp, u64, P(_Params64), P(_Grid64)]
    L.ss_result_is_f64.argtypes = [vp]
    L.ss_result_vertices_f64.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_particle_densities_f64.argtypes = [vp, P(vp), P(u64)]
    L.ss_result_grid_f64.argtypes = [vp, P(_Grid64)]
    L.ss_result_subdomain_grid_f64.argtypes = [vp, P(_Grid64), P(i32)]
    L.ss_result_levelset_box_f64.argtypes = [vp, P(C.c_int64), P(C.c_int64), vp]
    if L.ss_abi_version() != 6:
        raise ImportError("libsplashsurf_hip.so ABI version mismatch")
    _lib = L
    return L


class Parameters:
    """`splashsurf_lib::Parameters<f32>` (lib.rs:158-210), absolute units."""

    def __init__(self, particle_radius, compact_support_radius, cube_size, rest_density=1000.0,
                 iso_surface_threshold=0.6, particle_aabb=None, enable_multi_threading=True, enable_simd=True,
                 subdomain_grid=True, subdomain_num_cubes_per_dim=64, auto_disable=True,
                 global_neighborhood_list=False):
        # kept in double; converted to the Real type of the call (f32 or f64) in _c()
        self.particle_radius = float(particle_radius)
        self.rest_density = float(rest_density)
        self.compact_support_radius = float(compact_support_radius)
        self.cube_size = float(cube_size)
        self.iso_surface_threshold = float(iso_surface_threshold)
        self.particle_aabb = particle_aabb
        self.enable_multi_threading = bool(enable_multi_threading)
        self.enable_simd = int(enable_simd)  # 0 scalar, 1 (True) the reference's SIMD arithmetic, 2 the same with v_sqrt_f32 (splashsurf_hip.h)
        self.subdomain_grid = bool(subdomain_grid)
        self.subdomain_num_cubes_per_dim = int(subdomain_num_cubes_per_dim)
        self.auto_disable = bool(auto_disable)
        self.global_neighborhood_list = bool(global_neighborhood_list)

    @classmethod
    def new_relative(cls, particle_radius, relative_compact_support_radius, relative_cube_size, dtype=np.float32, **kw):
        """lib.rs:216-226 (products formed in the Real type `dtype`)"""
        t = np.dtype(dtype).type
        r = t(particle_radius)
        return cls(r, r * t(relative_compact_support_radius), r * t(relative_cube_size), **kw)

    def _c(self, f64=False):
        p = _Params64() if f64 else _Params()
        t = np.float64 if f64 else np.float32
        p.particle_radius = t(self.particle_radius)
        p.rest_density = t(self.rest_density)
        p.compact_support_radius = t(self.compact_support_radius)
        p.cube_size = t(self.cube_size)
        p.iso_surface_threshold = t(self.iso_surface_threshold)
        if self.particle_aabb is not None:
            p.has_particle_aabb = 1
            for d in range(3):
                p.aabb_min[d] = t(self.particle_aabb[0][d])
                p.aabb_max[d] = t(self.particle_aabb[1][d])
        p.enable_multi_threading = int(self.enable_multi_threading)
        p.enable_simd = int(self.enable_simd)
        p.decomposition = 1 if self.subdomain_grid else 0
        p.subdomain_num_cubes_per_dim = self.subdomain_num_cubes_per_dim
        p.auto_disable = int(self.auto_disable)
        p.global_neighborhood_list = int(self.global_neighborhood_list)
        return p


class Aabb3d:
    def __init__(self, mn, mx):
        self.min = np.asarray(mn) if isinstance(mn, np.ndarray) else np.array(mn, dtype=np.float32)
        self.max = np.asarray(mx) if isinstance(mx, np.ndarray) else np.array(mx, dtype=np.float32)


class UniformGrid:
    """`UniformGrid<i64, f32>` view (pysplashsurf.pyi UniformGrid)."""

    def __init__(self, g):
        t = np.float64 if isinstance(g, _Grid64) else np.float32
        self.aabb = Aabb3d(np.array(list(g.aabb_min), dtype=t), np.array(list(g.aabb_max), dtype=t))
        self.cell_size = t(g.cell_size)
        self.npoints_per_dim = [int(x) for x in g.n_points]
        self.ncells_per_dim = [int(x) for x in g.n_cells]


class TriMesh3d:
    def __init__(self, owner):
        self._owner = owner

    @property
    def vertices(self):
        return self._owner._vertices()

    @property
    def triangles(self):
        return self._owner._triangles()

    @property
    def triangles_u32(self):
        return self._owner._triangles_u32()


def sync_tensor_producer(t):
    """Stream-ordering contract of the C ABI for device pointers (include/splashsurf_hip.h, "Stream ordering"): the
    library runs on its own non-blocking HIP stream, so whatever produced a device buffer must have completed before
    the pointer is handed over.  For a torch CUDA tensor that is torch's current stream on the tensor's device."""
    if getattr(t, "is_cuda", False):
        import torch
        torch.cuda.current_stream(t.device).synchronize()


class Context:
    """Owns the HIP context/stream and reusable device buffers (the reference's thread pool + workspace)."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        h = C.c_void_p()
        st = self._lib.ss_context_create(int(device_id), C.byref(h))
        if st != 0:
            raise SplashsurfError(st, "ss_context_create failed (no usable HIP device %d?)" % device_id)
        self._h = h
        self.device_id = device_id
        self._results = weakref.WeakSet()  # live SurfaceReconstruction objects: freed before the context goes away

    def close(self):
        if getattr(self, "_h", None):
            for r in list(getattr(self, "_results", ())):
                r._free()
            self._lib.ss_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, st):
        msg = self._lib.ss_last_error(self._h)
        msg = msg.decode() if msg else ""
        if st == 1:
            raise GridConstructionError(st, msg)
        raise SplashsurfError(st, msg)

    def set_full_levelset(self, on=True):
        """SS_OPTION_FULL_LEVELSET: evaluate the level set completely everywhere (no early exit inside the fluid); needed before
        `SurfaceReconstruction.levelset_box` is used to look at values away from the surface."""
        self._lib.ss_context_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        st = self._lib.ss_context_set_option(self._h, 1, 1 if on else 0)
        if st != 0:
            self._raise(st)

    def set_two_pass(self, mode=-1):
        """SS_OPTION_SPLAT_TWO_PASS: -1 automatic (default), 0 never, 1 always certify sub-blocks inside the fluid before evaluating them."""
        self._lib.ss_context_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        st = self._lib.ss_context_set_option(self._h, 2, int(mode))
        if st != 0:
            self._raise(st)

    def measure_hbm_bandwidth(self, nbytes=2 << 30, repetitions=5):
        """ss_measure_hbm_bandwidth: (read GB/s, copy GB/s) this device sustains for float4 streams over `nbytes` per buffer."""
        rd, cp = C.c_double(), C.c_double()
        self._lib.ss_measure_hbm_bandwidth.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        st = self._lib.ss_measure_hbm_bandwidth(self._h, int(nbytes), int(repetitions), C.byref(rd), C.byref(cp))
        if st != 0:
            self._raise(st)
        return float(rd.value), float(cp.value)

    def set_stream(self, hip_stream_ptr):
        st = self._lib.ss_context_set_stream(self._h, C.c_void_p(hip_stream_ptr))
        if st != 0:
            self._raise(st)

    def _as_ptr(self, particles):
        """Accept a float32 (N,3) numpy array (host) or anything with `data_ptr()` (torch tensor, host or HBM)."""
        if hasattr(particles, "data_ptr"):
            dt = str(particles.dtype)
            if tuple(particles.shape[1:]) != (3,) or dt not in ("torch.float32", "torch.float64") or not particles.is_contiguous():
                raise TypeError("particles tensor must be contiguous float32/float64 of shape (N, 3)")
            if getattr(particles, "is_cuda", False) and particles.device.index != self.device_id:
                raise ValueError("particles live on cuda:%s but this Context drives device %d" % (particles.device.index, self.device_id))
            sync_tensor_producer(particles)
            return C.c_void_p(particles.data_ptr()), int(particles.shape[0]), particles, dt == "torch.float64"
        a = np.asarray(particles)
        if a.dtype not in (np.float32, np.float64):  # pysplashsurf/src/reconstruction.rs:187-206 rejects other dtypes as well
            raise TypeError("unsupported particle dtype %s (float32 and float64 are supported)" % a.dtype)
        if a.ndim != 2 or a.shape[1] != 3:
            raise ValueError("particles must have shape (N, 3)")
        a = np.ascontiguousarray(a)
        return C.c_void_p(a.ctypes.data), int(a.shape[0]), a, a.dtype == np.float64

    def reconstruct(self, particles, parameters, out=None):
        """dtype dispatch like pysplashsurf (reconstruction.rs:187-206): float32 -> <i64,f32>, float64 -> <i64,f64>."""
        ptr, n, keep, f64 = self._as_ptr(particles)
        p = parameters._c(f64)
        fn = self._lib.ss_reconstruct_surface_f64 if f64 else self._lib.ss_reconstruct_surface_f32
        fn_in = self._lib.ss_reconstruct_surface_inplace_f64 if f64 else self._lib.ss_reconstruct_surface_inplace_f32
        if out is None:
            h = C.c_void_p()
            st = fn(self._h, ptr, n, C.byref(p), C.byref(h))
            if st != 0:
                self._raise(st)
            return SurfaceReconstruction(self, h)
        st = fn_in(self._h, ptr, n, C.byref(p), out._h)
        if st != 0:
            self._raise(st)
        out._invalidate()
        return out

    def grid_for_reconstruction(self, particles, parameters):
        ptr, n, keep, f64 = self._as_ptr(particles)
        p = parameters._c(f64)
        g = _Grid64() if f64 else _Grid()
        fn = self._lib.ss_grid_for_reconstruction_f64 if f64 else self._lib.ss_grid_for_reconstruction_f32
        st = fn(self._h, ptr, n, C.byref(p), C.byref(g))
        if st != 0:
            self._raise(st)
        return UniformGrid(g)


class SurfaceReconstruction:
    """`SurfaceReconstruction<i64, f32>` (lib.rs:247-262)."""

    def __init__(self, ctx, handle):
        self._ctx = ctx
        self._lib = ctx._lib
        self._h = handle
        self._cache = {}
        self.mesh = TriMesh3d(self)
        ctx._results.add(self)

    def _invalidate(self):
        self._cache = {}

    def _free(self):
        if self._h:
            self._lib.ss_result_free(self._h)  # only needs hipSetDevice; valid as long as the context still exists
            self._h = None

    def __del__(self):
        try:
            if self._ctx._h:
                self._free()
            self._h = None
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            self._ctx._raise(st)

    def _host_array(self, fn, ctype, width, dtype, copy=True):
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(fn(self._h, C.byref(ptr), C.byref(n)))
        cnt = int(n.value) * width
        if cnt == 0 or not ptr.value:
            shape = (0, width) if width > 1 else (0,)
            return np.zeros(shape, dtype=dtype)
        arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(cnt,))
        if copy:
            arr = arr.copy()
        return arr.reshape(-1, width) if width > 1 else arr

    def mesh_views(self, u64=False):
        """(vertices, triangles_u32) as zero-copy numpy views of the library's pinned host buffers -- what a host
        caller of the C ABI gets from ss_result_vertices / ss_result_triangles_u32.  Valid only until this result is
        reused by another reconstruction or freed; use `.mesh.vertices` / `.mesh.triangles` for owning copies."""
        if self.is_f64:
            v = self._host_array(self._lib.ss_result_vertices_f64, C.c_double, 3, np.float64, copy=False)
        else:
            v = self._host_array(self._lib.ss_result_vertices, C.c_float, 3, np.float32, copy=False)
        if u64:  # the reference's index type ([usize; 3], lib.rs:247-262): widened on the device, 24 B per triangle over PCIe
            t = self._host_array(self._lib.ss_result_triangles, C.c_uint64, 3, np.uint64, copy=False)
        else:
            t = self._host_array(self._lib.ss_result_triangles_u32, C.c_uint32, 3, np.uint32, copy=False)
        return v, t

    def counts(self):
        nv, nt = C.c_uint64(), C.c_uint64()
        self._check(self._lib.ss_result_counts(self._h, C.byref(nv), C.byref(nt)))
        return int(nv.value), int(nt.value)

    @property
    def is_f64(self):
        return bool(self._lib.ss_result_is_f64(self._h))

    def _vertices(self):
        if "v" not in self._cache:
            if self.is_f64:
                self._cache["v"] = self._host_array(self._lib.ss_result_vertices_f64, C.c_double, 3, np.float64)
            else:
                self._cache["v"] = self._host_array(self._lib.ss_result_vertices, C.c_float, 3, np.float32)
        return self._cache["v"]

    def _triangles(self):
        if "t" not in self._cache:
            self._cache["t"] = self._host_array(self._lib.ss_result_triangles, C.c_uint64, 3, np.uint64)
        return self._cache["t"]

    def _triangles_u32(self):
        if "t32" not in self._cache:
            self._cache["t32"] = self._host_array(self._lib.ss_result_triangles_u32, C.c_uint32, 3, np.uint32)
        return self._cache["t32"]

    @property
    def vertex_keys(self):
        if "k" not in self._cache:
            self._cache["k"] = self._host_array(self._lib.ss_result_vertex_keys, C.c_uint64, 1, np.uint64)
        return self._cache["k"]

    @property
    def grid(self):
        f64 = self.is_f64
        g = _Grid64() if f64 else _Grid()
        self._check((self._lib.ss_result_grid_f64 if f64 else self._lib.ss_result_grid)(self._h, C.byref(g)))
        return UniformGrid(g)

    @property
    def subdomain_grid(self):
        f64 = self.is_f64
        g = _Grid64() if f64 else _Grid()
        present = C.c_int32()
        self._check((self._lib.ss_result_subdomain_grid_f64 if f64 else self._lib.ss_result_subdomain_grid)(self._h, C.byref(g), C.byref(present)))
        return UniformGrid(g) if present.value else None

    @property
    def particle_densities(self):
        if "rho" not in self._cache:
            if self.is_f64:
                self._cache["rho"] = self._host_array(self._lib.ss_result_particle_densities_f64, C.c_double, 1, np.float64)
            else:
                self._cache["rho"] = self._host_array(self._lib.ss_result_particle_densities, C.c_float, 1, np.float32)
        return self._cache["rho"]

    @property
    def particle_inside_aabb(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.ss_result_particle_inside_aabb(self._h, C.byref(ptr), C.byref(n)))
        if not ptr.value:
            return None
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(int(n.value),)).copy().astype(bool)

    @property
    def particle_neighbors_csr(self):
        """(row_ptr uint64[N+1], neighbors uint64[M]) or None when not requested."""
        if "nb" not in self._cache:
            rp, nb, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
            self._check(self._lib.ss_result_particle_neighbors(self._h, C.byref(rp), C.byref(nb), C.byref(n)))
            if not rp.value:
                self._cache["nb"] = None
            else:
                nrow = int(n.value) + 1
                row = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint64)), shape=(nrow,)).copy()
                m = int(row[-1])
                idx = np.ctypeslib.as_array(C.cast(nb, C.POINTER(C.c_uint64)), shape=(m,)).copy() if m else np.zeros(0, np.uint64)
                self._cache["nb"] = (row, idx)
        return self._cache["nb"]

    @property
    def particle_neighbors(self):
        """`SurfaceReconstruction::particle_neighbors`: list of per-particle neighbour index arrays (None if not requested)."""
        csr = self.particle_neighbors_csr
        if csr is None:
            return None
        row, idx = csr
        return [idx[int(row[i]):int(row[i + 1])] for i in range(row.size - 1)]

    @property
    def stats(self):
        s = _Stats()
        self._check(self._lib.ss_result_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _Stats._fields_}

    def levelset_box(self, lo, extent):
        if self.stats.get("n_truncated_blocks", 0):
            raise RuntimeError("this reconstruction stopped accumulating inside the fluid (values there are lower bounds); call "
                               "Context.set_full_levelset(True) before reconstructing to inspect the complete level set")
        lo_a = (C.c_int64 * 3)(*[int(x) for x in lo])
        ex_a = (C.c_int64 * 3)(*[int(x) for x in extent])
        f64 = self.is_f64
        out = np.zeros(tuple(int(x) for x in extent), dtype=np.float64 if f64 else np.float32)
        fn = self._lib.ss_result_levelset_box_f64 if f64 else self._lib.ss_result_levelset_box
        self._check(fn(self._h, lo_a, ex_a, out.ctypes.data_as(C.c_void_p)))
        return out

    def certified_subblocks(self):
        """ss_result_debug_certified (test aid): (masks[n_active] uint32, block_xyz[n_active, 3] uint32) -- the 4^3 sub-blocks the lower bound
        certified inside the fluid and the splat never evaluated."""
        n = C.c_uint64()
        fn = self._lib.ss_result_debug_certified
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        self._check(fn(self._h, None, None, 0, C.byref(n)))
        masks = np.zeros(int(n.value), dtype=np.uint32)
        xyz = np.zeros((int(n.value), 3), dtype=np.uint32)
        if n.value:
            self._check(fn(self._h, masks.ctypes.data_as(C.c_void_p), xyz.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return masks, xyz

    def subdomain_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._lib.ss_result_subdomain_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def device_pointers(self):
        out = {}
        for name, fn in (("vertices", self._lib.ss_result_device_vertices),
                         ("triangles_u32", self._lib.ss_result_device_triangles_u32),
                         ("particle_densities", self._lib.ss_result_device_particle_densities)):
            ptr, n = C.c_void_p(), C.c_uint64()
            self._check(fn(self._h, C.byref(ptr), C.byref(n)))
            out[name] = (ptr.value, int(n.value))
        return out


class _SlotContext:
    """The context of a pipeline slot as the accessors of `SurfaceReconstruction` see it (error messages); owned by the pipeline."""

    def __init__(self, lib, handle, device_id):
        self._lib = lib
        self._h = handle
        self.device_id = device_id
        self._results = weakref.WeakSet()

    _raise = Context._raise
    _as_ptr = Context._as_ptr


class _SlotResult(SurfaceReconstruction):
    """A result that belongs to a pipeline slot: never freed from Python."""

    def _free(self):
        self._h = None


class FramePipeline:
    """`ss_pipeline_*` (include/splashsurf_hip.h, csrc/ss_pipeline.hip): a time series of frames through `depth` contexts of one device -- the
    reference's loop of `reconstruct_surface_inplace` over the frames (lib.rs:340-346) with consecutive frames overlapping (upload and kernels of
    frame k + 1 beside the mesh download of frame k).  Frames come back in submission order::

        with FramePipeline(depth=2) as pipe:
            for mesh in pipe.map(frames, parameters):   # frames: iterable of (N, 3) float32 / float64 arrays
                use(mesh.mesh_views())                   # valid until `depth` further frames were submitted

    `fetch`: the host mirrors a slot's thread fills before its frame counts as complete."""

    FETCH_VERTICES, FETCH_TRIANGLES_U64, FETCH_TRIANGLES_U32, FETCH_DENSITIES = 1, 2, 4, 8

    def __init__(self, device_id=0, depth=2):
        self._lib = L = load_library()
        vp, u64 = C.c_void_p, C.c_uint64
        L.ss_pipeline_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        L.ss_pipeline_destroy.argtypes = [vp]
        L.ss_pipeline_destroy.restype = None
        L.ss_pipeline_last_error.argtypes = [vp]
        L.ss_pipeline_last_error.restype = C.c_char_p
        L.ss_pipeline_depth.argtypes = [vp]
        L.ss_pipeline_in_flight.argtypes = [vp]
        L.ss_pipeline_context.argtypes = [vp, C.c_int]
        L.ss_pipeline_context.restype = vp
        L.ss_pipeline_set_option.argtypes = [vp, C.c_int, C.c_int]
        L.ss_pipeline_submit_f32.argtypes = [vp, vp, u64, C.POINTER(_Params), C.c_uint32, C.POINTER(u64)]
        L.ss_pipeline_submit_f64.argtypes = [vp, vp, u64, C.POINTER(_Params64), C.c_uint32, C.POINTER(u64)]
        L.ss_pipeline_next.argtypes = [vp, C.POINTER(vp), C.POINTER(u64)]
        L.ss_pipeline_ready.argtypes = [vp]
        L.ss_pipeline_frame_times.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        h = vp()
        st = L.ss_pipeline_create(int(device_id), int(depth), C.byref(h))
        if st != 0:
            raise SplashsurfError(st, "ss_pipeline_create failed (device %d, depth %d)" % (device_id, depth))
        self._h = h
        self.device_id = device_id
        self.depth = int(depth)
        self._slot_ctx = [_SlotContext(L, vp(L.ss_pipeline_context(h, i)), device_id) for i in range(self.depth)]
        self._slot_res = {}   # result handle -> _SlotResult
        self._keep = {}       # ticket -> the particle array of a frame in flight (the library reads it until the frame is handed back)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ss_pipeline_destroy(self._h)
            self._h = None
            for r in self._slot_res.values():
                r._h = None
            for c in self._slot_ctx:
                c._h = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _raise(self, st):
        msg = self._lib.ss_pipeline_last_error(self._h)
        msg = msg.decode() if msg else ""
        if st == 1:
            raise GridConstructionError(st, msg)
        raise SplashsurfError(st, msg)

    def set_option(self, option, value):
        st = self._lib.ss_pipeline_set_option(self._h, int(option), int(value))
        if st != 0:
            self._raise(st)

    def set_two_pass(self, mode=-1):
        self.set_option(2, mode)

    @property
    def in_flight(self):
        return int(self._lib.ss_pipeline_in_flight(self._h))

    def ready(self):
        return bool(self._lib.ss_pipeline_ready(self._h))

    def submit(self, particles, parameters, fetch=FETCH_VERTICES | FETCH_TRIANGLES_U32):
        """Queues a frame; returns its ticket.  The array is kept alive (and must stay unchanged) until the frame was handed back."""
        ptr, n, keep, f64 = self._slot_ctx[0]._as_ptr(particles)
        p = parameters._c(f64)
        t = C.c_uint64()
        fn = self._lib.ss_pipeline_submit_f64 if f64 else self._lib.ss_pipeline_submit_f32
        st = fn(self._h, ptr, n, C.byref(p), int(fetch), C.byref(t))
        if st != 0:
            self._raise(st)
        self._keep[int(t.value)] = keep
        return int(t.value)

    def next(self):
        """Blocks for the oldest frame in flight: (ticket, SurfaceReconstruction).  The result is valid until its slot is reused (the `depth`-th
        submit after the frame's own); a failed frame raises like `Context.reconstruct` does."""
        h, t = C.c_void_p(), C.c_uint64()
        st = self._lib.ss_pipeline_next(self._h, C.byref(h), C.byref(t))
        self._keep.pop(int(t.value), None)
        if st != 0:
            self._raise(st)
        r = self._slot_res.get(h.value)
        if r is None:
            r = self._slot_res[h.value] = _SlotResult(self._slot_ctx[int(t.value) % self.depth], h)
        r._invalidate()
        return int(t.value), r

    def frame_times(self, slot):
        a, b = C.c_double(), C.c_double()
        st = self._lib.ss_pipeline_frame_times(self._h, int(slot), C.byref(a), C.byref(b))
        if st != 0:
            self._raise(st)
        return float(a.value), float(b.value)

    def map(self, frames, parameters, fetch=FETCH_VERTICES | FETCH_TRIANGLES_U32):
        """Generator over the results of `frames` (an iterable of particle arrays), in order, `depth` frames in flight."""
        it = iter(frames)
        done = False
        while True:
            while not done and self.in_flight < self.depth:
                try:
                    self.submit(next(it), parameters, fetch)
                except StopIteration:
                    done = True
            if self.in_flight == 0:
                return
            yield self.next()[1]


_default_ctx = {}


def default_context(device_id=0):
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]


def reconstruct_surface_abs(particles, parameters, context=None, out=None):
    """`splashsurf_lib::reconstruct_surface::<i64, f32>(positions, &parameters)` (lib.rs:330-337)."""
    ctx = context or default_context()
    return ctx.reconstruct(particles, parameters, out=out)


def reconstruct_surface(particles, *, particle_radius, rest_density=1000.0, smoothing_length, cube_size,
                        iso_surface_threshold=0.6, aabb_min=None, aabb_max=None, multi_threading=True,
                        simd=True, global_neighborhood_list=False, subdomain_grid=True,
                        subdomain_grid_auto_disable=True, subdomain_num_cubes_per_dim=64, context=None):
    """Signature of `pysplashsurf.reconstruct_surface` (pysplashsurf/src/reconstruction.rs:135-207).

    `smoothing_length` and `cube_size` are relative to `particle_radius`:
    compact_support_radius = 2 * smoothing_length * particle_radius, cube = cube_size * particle_radius,
    both formed in f64 and then cast to f32 (reconstruction.rs:171-193).
    """
    aabb = None
    if aabb_min is not None and aabb_max is not None:
        aabb = (np.asarray(aabb_min, dtype=np.float64), np.asarray(aabb_max, dtype=np.float64))
    r = float(particle_radius)
    prm = Parameters(
        particle_radius=r, rest_density=float(rest_density),
        compact_support_radius=2.0 * float(smoothing_length) * r,   # f64 products, cast to the Real type of
        cube_size=float(cube_size) * r,                              # the call (reconstruction.rs:171-193)
        iso_surface_threshold=float(iso_surface_threshold), particle_aabb=aabb,
        enable_multi_threading=multi_threading, enable_simd=simd, subdomain_grid=subdomain_grid,
        subdomain_num_cubes_per_dim=subdomain_num_cubes_per_dim, auto_disable=subdomain_grid_auto_disable,
        global_neighborhood_list=global_neighborhood_list)
    return reconstruct_surface_abs(particles, prm, context=context)


def _new_result(ctx):
    h = C.c_void_p()
    st = ctx._lib.ss_result_create(ctx._h, C.byref(h))
    if st != 0:
        ctx._raise(st)
    return SurfaceReconstruction(ctx, h)


def marching_cubes(values, *, iso_surface_threshold, cube_size, translation=None, return_grid=False, context=None):
    """`pysplashsurf.marching_cubes` (pysplashsurf/src/marching_cubes.rs:58-127): marching cubes on a dense 3D array of
    function values at the points translation + (i, j, k) * cube_size; the reference's global-strategy triangulation
    (marching_cubes.rs:100-127).  Returns the TriMesh3d (vertices in ascending edge-key order), optionally with the grid."""
    ctx = context or default_context()
    L = ctx._lib
    if hasattr(values, "data_ptr"):  # torch tensor (host or HBM)
        vals = values.contiguous()
        if str(vals.dtype) not in ("torch.float32", "torch.float64"):
            raise TypeError("values must be float32 or float64")
        f64 = str(vals.dtype) == "torch.float64"
        ptr = C.c_void_p(vals.data_ptr())
    else:
        vals = np.asarray(values)
        if vals.dtype not in (np.float32, np.float64):
            raise TypeError("values must be float32 or float64")
        vals = np.ascontiguousarray(vals)
        f64 = vals.dtype == np.float64
        ptr = C.c_void_p(vals.ctypes.data)
    if len(vals.shape) != 3:
        raise ValueError("values must be a 3D array")
    real = C.c_double if f64 else C.c_float
    dt = np.float64 if f64 else np.float32
    npts = (C.c_int64 * 3)(*[int(x) for x in vals.shape])
    tr = None if translation is None else (real * 3)(*[float(dt(x)) for x in translation])
    res = _new_result(ctx)
    fn = L.ss_marching_cubes_f64 if f64 else L.ss_marching_cubes_f32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), real, real, C.c_void_p, C.c_void_p]
    st = fn(ctx._h, ptr, npts, real(float(dt(iso_surface_threshold))), real(float(dt(cube_size))), tr, res._h)
    if st != 0:
        ctx._raise(st)
    return (res.mesh, res.grid) if return_grid else res.mesh


class NeighborhoodLists:
    """pysplashsurf.NeighborhoodLists: per-particle neighbour index lists."""

    def __init__(self, result):
        self._result = result

    @property
    def csr(self):
        return self._result.particle_neighbors_csr

    def get_neighborhood_lists(self):
        return self._result.particle_neighbors


def neighborhood_search_spatial_hashing_parallel(particle_positions, domain, search_radius, context=None):
    """`pysplashsurf.neighborhood_search_spatial_hashing_parallel(particle_positions, domain, search_radius)`; lists come
    in the order of the reference's sequential function (neighborhood_search.rs:131-230) -- the parallel one's per-cell
    order depends on thread timing."""
    ctx = context or default_context()
    L = ctx._lib
    ptr, n, keep, f64 = ctx._as_ptr(particle_positions)
    real = C.c_double if f64 else C.c_float
    dt = np.float64 if f64 else np.float32
    lo = (real * 3)(*[float(dt(x)) for x in domain.min])
    hi = (real * 3)(*[float(dt(x)) for x in domain.max])
    res = _new_result(ctx)
    fn = L.ss_neighborhood_search_f64 if f64 else L.ss_neighborhood_search_f32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(real), C.POINTER(real), real, C.c_void_p]
    st = fn(ctx._h, ptr, n, lo, hi, real(float(dt(search_radius))), res._h)
    if st != 0:
        ctx._raise(st)
    return NeighborhoodLists(res)


def grid_for_reconstruction(particles, parameters, context=None):
    """`splashsurf_lib::grid_for_reconstruction` (lib.rs:476-516)."""
    ctx = context or default_context()
    return ctx.grid_for_reconstruction(particles, parameters)
