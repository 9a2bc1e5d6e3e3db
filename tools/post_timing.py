"""Stage timings of the on-device post-processing on a large mesh (HBM-resident tensors, C ABI called in place)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "/root/repo")
import splashsurf_amd as S
from splashsurf_amd import workloads as W, postprocessing as PP
from splashsurf_amd.api import Context, Parameters
import ctypes as C

wl = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "s10m_tank"]
r = wl["particle_radius"]
ctx = Context(0)
pts = wl["gen"]()
d_pts = torch.from_numpy(pts).cuda()
prm = Parameters(particle_radius=r, compact_support_radius=np.float32(2.0 * wl["smoothing_length"] * r), cube_size=np.float32(wl["cube_size"] * r),
                 auto_disable=False, global_neighborhood_list=True)
h = np.float32(2.0 * wl["smoothing_length"] * r)
def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print("%-28s %8.3f ms" % (name, (time.perf_counter() - t) / reps * 1e3), flush=True)
    return out
rec = timed("reconstruct (+nbr lists)", lambda: ctx.reconstruct(d_pts, prm), reps=2)
nv, nt = rec.counts()
print("particles", pts.shape[0], "vertices", nv, "triangles", nt)
L = PP._lib()
d_v = torch.empty((nv, 3), dtype=torch.float32, device="cuda"); d_t = torch.empty((nt, 3), dtype=torch.int32, device="cuda")
d_rho = torch.empty((pts.shape[0],), dtype=torch.float32, device="cuda")
L.ss_result_copy_vertices(rec._h, C.c_void_p(d_v.data_ptr())); L.ss_result_copy_triangles_u32(rec._h, C.c_void_p(d_t.data_ptr()))
L.ss_result_copy_particle_densities(rec._h, C.c_void_p(d_rho.data_ptr()))
conn = timed("vertex connectivity", lambda: PP.vertex_vertex_connectivity(nv, d_t, ctx))
row, idx, n_p, n_e = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
L.ss_result_device_particle_neighbors(rec._h, C.byref(row), C.byref(idx), C.byref(n_p), C.byref(n_e))
print("neighbour entries", n_e.value)
wnc = timed("weighted neighbour counts", lambda: PP.weighted_neighbor_counts(d_pts, row, idx, h, ctx))
rr = np.float32(r); mass = np.float32(4.0) * np.float32(np.pi / 3) * (rr * rr * rr) * np.float32(1000.0)
interp = PP.SphInterpolator(d_pts, d_rho, mass, h, context=ctx)
wnn = timed("SPH interpolate wnn (1st ord)", lambda: interp.interpolate_quantity(wnc, d_v, first_order_correction=True))
sw = timed("smooth-step weights", lambda: PP.smoothing_weights(wnn, 13.0, ctx))
mesh = PP.TriMesh3d(d_v, d_t, ctx)
timed("laplacian smoothing x25", lambda: PP.laplacian_smoothing_parallel(mesh, conn, iterations=25, beta=1.0, weights=sw), reps=2)
n = timed("vertex normals", lambda: PP.vertex_normals(d_v, d_t, ctx))
timed("SPH normals", lambda: interp.interpolate_normals(d_v))
timed("normal smoothing x10", lambda: PP.laplacian_smoothing_normals_parallel(n, conn, iterations=10, context=ctx), reps=2)
