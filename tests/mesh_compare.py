"""Order-independent mesh comparison helpers used by the parity tests.

Two notions of vertex identity are supported:

* *native keys*: both the CPU oracle and the HIP path report, per vertex, the global grid edge
  carrying it (key = ((gi*NPy+gj)*NPz+gk)*3 + axis).  With keys the comparison is exact.
* *geometric cluster ids* (for meshes that come without keys, i.e. the reference's own output stored
  in tests/golden/): every marching-cubes vertex lies on a grid edge, so its edge can be recovered
  from the coordinates.  Vertices that sit (numerically) ON a grid point -- interpolation weight
  alpha ~ 0 or 1 -- are ambiguous between the incident edges and are clustered per grid point.

Triangles are compared as multisets of rotation-normalised index triplets (orientation preserved),
which is what "triangle index sets identical modulo ordering" means for this path.
"""
import numpy as np


def canonical_triangles(tri):
    """Rotate each triangle so its smallest id comes first (orientation kept), then sort rows."""
    t = np.asarray(tri).astype(np.int64).reshape(-1, 3)
    if t.shape[0] == 0:
        return t
    amin = np.argmin(t, axis=1)
    rows = np.arange(t.shape[0])
    t = np.stack([t[rows, amin], t[rows, (amin + 1) % 3], t[rows, (amin + 2) % 3]], axis=1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


def canonicalize_keyed(vertices, keys, triangles):
    """Sort vertices by native key; remap + canonicalise triangles."""
    keys = np.asarray(keys).astype(np.uint64)
    order = np.argsort(keys, kind="stable")
    inv = np.empty(order.size, dtype=np.int64)
    inv[order] = np.arange(order.size)
    tri = np.asarray(triangles).astype(np.int64).reshape(-1, 3)
    t = canonical_triangles(inv[tri]) if tri.size else tri
    return np.asarray(vertices)[order], keys[order], t


def geometric_cluster_ids(vertices, grid_min, cell_size, n_points, tol=1e-3):
    """Recover (edge | grid point) ids from coordinates.  Returns int64 ids:
    id = ((i*NPy+j)*NPz+k)*4 + axis   for a vertex strictly inside an edge (axis 0..2),
    id = ((i*NPy+j)*NPz+k)*4 + 3      for a vertex sitting on grid point (i,j,k)."""
    v = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    gmin = np.asarray(grid_min, dtype=np.float64)
    u = (v - gmin) / float(cell_size)
    r = np.rint(u)
    # tolerance in cell units: `tol` plus a few f32 ulps of the coordinate (matters far from the origin)
    tol_u = tol + 6.0 * 1.1920929e-07 * np.maximum(np.abs(v), np.abs(gmin)) / float(cell_size)
    on_line = np.abs(u - r) < tol_u
    n_free = (~on_line).sum(axis=1)
    if np.any(n_free > 1):
        raise AssertionError("vertex not on a grid edge: %d offenders" % int((n_free > 1).sum()))
    axis = np.where(n_free == 1, np.argmax(~on_line, axis=1), 3)
    ijk = r.astype(np.int64)
    fl = np.floor(u).astype(np.int64)
    rows = np.arange(v.shape[0])
    m = axis < 3
    ijk[rows[m], axis[m]] = fl[rows[m], axis[m]]
    npx, npy, npz = [int(x) for x in n_points]
    return ((ijk[:, 0] * npy + ijk[:, 1]) * npz + ijk[:, 2]) * 4 + axis


def canonicalize_geometric(vertices, triangles, grid_min, cell_size, n_points, tol=1e-3):
    """Returns (sorted cluster ids, vertices sorted alike, canonical triangles over cluster ids)."""
    ids = geometric_cluster_ids(vertices, grid_min, cell_size, n_points, tol)
    order = np.argsort(ids, kind="stable")
    tri = np.asarray(triangles).astype(np.int64).reshape(-1, 3)
    t = canonical_triangles(ids[tri]) if tri.size else tri
    return ids[order], np.asarray(vertices)[order], t


def compare_keyed(va, ka, ta, vb, kb, tb):
    """Exact comparison of two keyed meshes. Returns a dict of findings."""
    va, ka, ta = canonicalize_keyed(va, ka, ta)
    vb, kb, tb = canonicalize_keyed(vb, kb, tb)
    out = dict(n_vertices=(len(ka), len(kb)), n_triangles=(len(ta), len(tb)))
    out["keys_equal"] = bool(np.array_equal(ka, kb))
    out["triangles_equal"] = bool(ta.shape == tb.shape and np.array_equal(ta, tb))
    if out["keys_equal"] and len(ka):
        both64 = np.asarray(va).dtype == np.float64 and np.asarray(vb).dtype == np.float64
        dt, U = (np.float64, np.uint64) if both64 else (np.float32, np.uint32)  # f64 meshes are compared as 64-bit patterns
        a32 = np.ascontiguousarray(va, dtype=dt)
        b32 = np.ascontiguousarray(vb, dtype=dt)
        out["vertices_bit_equal"] = bool(np.array_equal(a32.view(U), b32.view(U)))
        out["n_vertices_differing"] = int(np.any(a32 != b32, axis=1).sum())
        out["max_abs_diff"] = float(np.max(np.abs(a32.astype(np.float64) - b32.astype(np.float64))))
    else:
        out["vertices_bit_equal"] = bool(out["keys_equal"])
        out["n_vertices_differing"] = 0
        out["max_abs_diff"] = 0.0
    return out


def compare_geometric(va, ta, vb, tb, grid_min, cell_size, n_points, tol=1e-3):
    """Comparison of two key-less meshes via geometric cluster ids."""
    ia, va_s, tca = canonicalize_geometric(va, ta, grid_min, cell_size, n_points, tol)
    ib, vb_s, tcb = canonicalize_geometric(vb, tb, grid_min, cell_size, n_points, tol)
    out = dict(n_vertices=(len(ia), len(ib)), n_triangles=(len(tca), len(tcb)))
    out["ids_equal"] = bool(np.array_equal(ia, ib))
    out["triangles_equal"] = bool(tca.shape == tcb.shape and np.array_equal(tca, tcb))
    if out["ids_equal"] and len(ia):
        a = va_s.astype(np.float64)
        b = vb_s.astype(np.float64)
        diff = np.max(np.abs(a - b), axis=1)
        # Clusters with several vertices (all within ~tol*cell_size of one grid point) come in arbitrary
        # order: pair every vertex with the nearest vertex of the same cluster in the other mesh
        # (symmetric Hausdorff distance inside the cluster).
        same_prev = np.concatenate([[False], ia[1:] == ia[:-1]])
        same_next = np.concatenate([ia[1:] == ia[:-1], [False]])
        multi = same_prev | same_next
        if multi.any():
            idx = np.nonzero(multi)[0]
            starts = idx[~same_prev[idx]]
            for s0 in starts:
                e0 = s0 + 1
                while e0 < len(ia) and ia[e0] == ia[s0]:
                    e0 += 1
                A, B = a[s0:e0], b[s0:e0]
                d = np.max(np.abs(A[:, None, :] - B[None, :, :]), axis=2)
                diff[s0:e0] = np.maximum(d.min(axis=1), d.min(axis=0))
        out["max_abs_diff"] = float(diff.max())
        scale = np.maximum(np.max(np.abs(a), axis=1), 1e-30)
        out["max_rel_diff"] = float(np.max(diff / scale))
        out["n_vertices_bit_equal"] = int(np.all(va_s.astype(np.float32) == vb_s.astype(np.float32), axis=1).sum())
        out["n_multi_cluster_vertices"] = int(multi.sum())
    elif out["ids_equal"]:
        out["max_abs_diff"] = 0.0
        out["max_rel_diff"] = 0.0
        out["n_vertices_bit_equal"] = 0
    else:
        out["max_abs_diff"] = float("nan")
        out["max_rel_diff"] = float("nan")
        out["n_vertices_bit_equal"] = 0
    return out


def mesh_is_closed_manifold(triangles):
    """Every directed edge appears exactly once and its reverse exactly once (closed, oriented
    2-manifold in the edge sense) -- the property the reference asserts through
    check_mesh_consistency (marching_cubes.rs:129-213) in test_full.rs:144-157."""
    t = np.asarray(triangles).astype(np.int64).reshape(-1, 3)
    if t.shape[0] == 0:
        return True
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]], axis=0)
    nv = int(t.max()) + 1
    fwd = e[:, 0] * nv + e[:, 1]
    rev = e[:, 1] * nv + e[:, 0]
    uf, cf = np.unique(fwd, return_counts=True)
    if np.any(cf != 1):
        return False
    return bool(np.array_equal(uf, np.unique(rev)))
