"""Multi-GPU surface reconstruction, host mirror: one process per GPU over torch.distributed (backend "nccl" = RCCL
over xGMI on a GPU node, "gloo" in the CPU tests).

The reference is single-process; its only natural decomposition is the uniform grid of subdomains
(dense_subdomains.rs:349-494 builds it, :1582-1598 iterates it in parallel).  Here the subdomain
grid of ONE global domain is cut into `world` axis-aligned BRICKS of subdomains by recursive coordinate bisection over
the histogram of owner subdomains, i.e. balanced by particle count.  One sharded reconstruction ("step"):

  1. global particle ids = concatenation of the ranks' inputs (the summation order of the level set is "ascending
     particle index", so ids must be global); global AABB by all-reduce(MIN/MAX);
  2. every rank derives the same global grid; the 3-D histogram of owner subdomains is all-reduced and bisected
     (`bricks_from_histogram`), so every rank holds the same partition;
  3. sparse all-to-all #1, one message per neighbour: (id, position) of every particle inside the neighbour's brick
     grown by the ghost margin -- only halo layers travel once the input is roughly where it belongs;
  4. phase 1 on each rank: binning + densities of the particles CONTAINED in its brick (ss_shard_begin);
  5. sparse all-to-all #2: owners send (id, rho) to the ranks that hold the particle as a ghost (values are copied,
     never recomputed, so densities are bit-identical to a single-process run);
  6. phase 2: level set + marching cubes of the brick (ss_shard_finish).  Vertices on brick faces are produced by every
     adjacent rank with identical global edge keys and identical coordinates;
  7. assembly (`assemble`): a face vertex belongs to the LOWEST rank whose brick holds its edge (the rule of
     globalize_local_edge, dense_subdomains.rs:1260-1329: the lower-side patch owns a boundary edge); counts are
     all-gathered into global vertex / triangle offsets, owners tell the other ranks the global ids of shared
     vertices (sparse all-to-all #3, keyed by the global edge key -- the hash join of `stitching`,
     dense_subdomains.rs:1693-1733, as a sort + binary search), triangles are rewritten to global ids.  The mesh is
     then the concatenation over ranks of (owned vertices, triangles); nothing is de-duplicated after the fact.

The per-rank engine is pluggable: `HipEngine` drives the C ABI (ss_shard_begin_f32 / ss_shard_finish), the CPU tests
plug in the oracle (tests/test_distributed.py).  The same algorithm runs natively (RCCL inside the library, no Python on
the data path) behind `ss_dist_reconstruct_*` (include/splashsurf_hip.h); this module is its host-side mirror and the
transport of the CPU tests.

A failing point-to-point exchange RAISES.  `SPLASH_EXCHANGE=allgather` selects the padded all-gather transport
explicitly (more bytes on the wire, only the most basic collective); nothing falls back silently.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist


class ShardDesc:
    def __init__(self, domain_min, domain_max, sub_lo, sub_hi):
        dt = np.float64 if np.asarray(domain_min).dtype == np.float64 else np.float32
        self.domain_min = np.asarray(domain_min, dtype=dt)
        self.domain_max = np.asarray(domain_max, dtype=dt)
        self.sub_lo = [int(x) for x in sub_lo]
        self.sub_hi = [int(x) for x in sub_hi]


class _Shard(C.Structure):
    _fields_ = [("domain_min", C.c_float * 3), ("domain_max", C.c_float * 3), ("sub_lo", C.c_int64 * 3), ("sub_hi", C.c_int64 * 3)]


class _Shard64(C.Structure):
    _fields_ = [("domain_min", C.c_double * 3), ("domain_max", C.c_double * 3), ("sub_lo", C.c_int64 * 3), ("sub_hi", C.c_int64 * 3)]


class HipEngine:
    """Per-rank engine on the HIP library (device tensors in, device-resident mesh out)."""

    def __init__(self, ctx, params, dtype=np.float32):
        from . import api
        self.api = api
        self.ctx = ctx
        self.params = params
        self.f64 = np.dtype(dtype) == np.float64  # Real type of the job: f32 -> ss_shard_*_f32, f64 -> *_f64
        self.np_dtype = np.float64 if self.f64 else np.float32
        self.torch_dtype = torch.float64 if self.f64 else torch.float32
        self.lib = ctx._lib
        vp, u64 = C.c_void_p, C.c_uint64
        self.lib.ss_shard_begin_f32.argtypes = [vp, vp, u64, C.POINTER(api._Params), C.POINTER(_Shard), vp]
        self.lib.ss_shard_begin_f64.argtypes = [vp, vp, u64, C.POINTER(api._Params64), C.POINTER(_Shard64), vp]
        self.lib.ss_shard_finish.argtypes = [vp, vp]
        for name in ("ss_shard_get_densities", "ss_shard_set_densities", "ss_shard_get_densities_f64", "ss_shard_set_densities_f64"):
            getattr(self.lib, name).argtypes = [vp, vp, u64]
        for name in ("ss_result_copy_vertices", "ss_result_copy_triangles_u32", "ss_result_copy_vertex_keys"):
            getattr(self.lib, name).argtypes = [vp, vp]
        self.lib.ss_grid_for_domain_f32.argtypes = [C.POINTER(api._Params), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(api._Grid),
                                                    C.POINTER(api._Grid), C.POINTER(C.c_float)]
        self.lib.ss_grid_for_domain_f64.argtypes = [C.POINTER(api._Params64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(api._Grid64),
                                                    C.POINTER(api._Grid64), C.POINTER(C.c_double)]
        h = C.c_void_p()
        st = self.lib.ss_result_create(ctx._h, C.byref(h))
        if st != 0:
            ctx._raise(st)
        self.result = api.SurfaceReconstruction(ctx, h)

    def grid_for_domain(self, dmin, dmax):
        p = self.params._c(self.f64)
        G, creal = (self.api._Grid64, C.c_double) if self.f64 else (self.api._Grid, C.c_float)
        g, sg, m = G(), G(), creal()
        a = (creal * 3)(*[float(x) for x in dmin])
        b = (creal * 3)(*[float(x) for x in dmax])
        fn = self.lib.ss_grid_for_domain_f64 if self.f64 else self.lib.ss_grid_for_domain_f32
        st = fn(C.byref(p), a, b, C.byref(g), C.byref(sg), C.byref(m))
        if st != 0:
            raise RuntimeError("ss_grid_for_domain failed: %d" % st)
        return (np.array(list(g.aabb_min), self.np_dtype), float(sg.cell_size), [int(x) for x in sg.n_cells], float(m.value),
                int(self.params.subdomain_num_cubes_per_dim))

    def _shard(self, sd):
        s = _Shard64() if self.f64 else _Shard()
        for d in range(3):
            s.domain_min[d] = float(sd.domain_min[d])
            s.domain_max[d] = float(sd.domain_max[d])
            s.sub_lo[d] = sd.sub_lo[d]
            s.sub_hi[d] = sd.sub_hi[d]
        return s

    def begin(self, local_pts, shard):
        """local_pts: contiguous (n,3) tensor of the job's Real type on this rank's device.  Returns densities (owned computed, others 0)."""
        assert local_pts.dtype == self.torch_dtype
        self.api.sync_tensor_producer(local_pts)
        p = self.params._c(self.f64)
        s = self._shard(shard)
        n = int(local_pts.shape[0])
        fn = self.lib.ss_shard_begin_f64 if self.f64 else self.lib.ss_shard_begin_f32
        st = fn(self.ctx._h, C.c_void_p(local_pts.data_ptr()), n, C.byref(p), C.byref(s), self.result._h)
        if st != 0:
            self.ctx._raise(st)
        self.result._invalidate()
        rho = torch.empty(n, dtype=self.torch_dtype, device=local_pts.device)
        fn = self.lib.ss_shard_get_densities_f64 if self.f64 else self.lib.ss_shard_get_densities
        st = fn(self.result._h, C.c_void_p(rho.data_ptr()), n)
        if st != 0:
            self.ctx._raise(st)
        return rho

    def finish(self, rho):
        self.api.sync_tensor_producer(rho)
        fn = self.lib.ss_shard_set_densities_f64 if self.f64 else self.lib.ss_shard_set_densities
        st = fn(self.result._h, C.c_void_p(rho.data_ptr()), int(rho.shape[0]))
        if st != 0:
            self.ctx._raise(st)
        st = self.lib.ss_shard_finish(self.ctx._h, self.result._h)
        if st != 0:
            self.ctx._raise(st)
        self.result._invalidate()
        return self.result

    def device_mesh(self, device):
        """(vertices (V,3) Real, keys (V,) int64, triangles (T,3) int64) of the last finish() as tensors on `device`; the
        mesh does not leave HBM when `device` is this context's GPU."""
        nv, nt = self.result.counts()
        v = torch.empty((nv, 3), dtype=self.torch_dtype, device=device)
        k = torch.empty((nv,), dtype=torch.int64, device=device)
        t32 = torch.empty((nt, 3), dtype=torch.int32, device=device)
        for fn, buf in ((self.lib.ss_result_copy_vertices, v), (self.lib.ss_result_copy_vertex_keys, k), (self.lib.ss_result_copy_triangles_u32, t32)):
            if buf.numel():
                st = fn(self.result._h, C.c_void_p(buf.data_ptr()))
                if st != 0:
                    self.ctx._raise(st)
        return v, k, t32.to(torch.int64) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------------------------------
# partition: recursive coordinate bisection of the subdomain grid, balanced by owner-particle counts
# ---------------------------------------------------------------------------------------------------------------------
def bricks_from_histogram(hist3, world, tol=0.02, axis_pref=(0.0, 0.0, 0.0)):
    """Cut the subdomain grid (shape ns of `hist3`, owner-particle count per subdomain) into `world` axis-aligned bricks
    [lo, hi) by recursive bisection: a box that has to serve k ranks is cut into parts for k//2 and k - k//2 ranks at the
    whole-subdomain plane that splits its particles closest to that ratio; among the axes whose best cut is within `tol`
    (fraction of the box's particles) of the best one, the cut with the smallest area (least halo) wins, then the axis with
    the smallest `axis_pref` (the caller passes how far the ranks' inputs extend along each axis: cutting where the inputs
    are already separated moves the fewest particles).  Pure integer /
    float64 arithmetic on identical inputs, so every rank derives the same partition.  Ranks that cannot get a subdomain
    (more ranks than subdomains in their box) receive empty bricks (lo == hi on one axis)."""
    hist3 = np.asarray(hist3, dtype=np.float64)
    ns = hist3.shape
    out = [None] * world

    def split(lo, hi, r0, r1):
        k = r1 - r0
        if k == 1:
            out[r0] = (tuple(lo), tuple(hi))
            return
        k1 = k // 2
        frac = k1 / k
        box = hist3[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        total = float(box.sum())
        ext = [hi[d] - lo[d] for d in range(3)]
        cands = []
        for a in range(3):
            if ext[a] < 2:
                continue
            marg = box.sum(axis=tuple(d for d in range(3) if d != a))
            cum = np.concatenate([[0.0], np.cumsum(marg)])
            if total > 0:
                c = int(np.argmin(np.abs(cum[1:ext[a]] - total * frac))) + 1
                err = abs(cum[c] - total * frac) / total
            else:
                c = min(max(int(np.floor(ext[a] * frac + 0.5)), 1), ext[a] - 1)
                err = 0.0
            area = 1
            for d in range(3):
                if d != a:
                    area *= ext[d]
            cands.append((err, area, a, c))
        if not cands:  # a single subdomain for several ranks: the first gets it, the others get empty bricks
            out[r0] = (tuple(lo), tuple(hi))
            for r in range(r0 + 1, r1):
                out[r] = (tuple(hi[:1]) + tuple(lo[1:]), tuple(hi))
            return
        best = min(c[0] for c in cands)
        err, area, a, c = min((x for x in cands if x[0] <= best + tol), key=lambda x: (x[1], axis_pref[x[2]], x[0], x[2]))
        mid_hi = list(hi)
        mid_hi[a] = lo[a] + c
        mid_lo = list(lo)
        mid_lo[a] = lo[a] + c
        split(list(lo), mid_hi, r0, r0 + k1)
        split(mid_lo, list(hi), r0 + k1, r1)

    split([0, 0, 0], list(ns), 0, world)
    return out


def slabs_from_histogram(hist, world):
    """1-D special case (kept for callers that want slabs along one axis): contiguous ranges of layers balanced by count."""
    ns_axis = int(len(hist))
    cum = np.concatenate([[0.0], np.cumsum(np.asarray(hist, dtype=np.float64))])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, ns_axis)] - target):
            k -= 1
        k = max(k, bounds[-1])
        k = min(k, ns_axis)
        bounds.append(k)
    bounds.append(ns_axis)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def partition_slabs(coords_axis, gmin_axis, sub_size, ns_axis, world):
    s = torch.floor((coords_axis - gmin_axis) / sub_size).to(torch.int64).clamp_(0, ns_axis - 1)
    hist = torch.bincount(s, minlength=ns_axis).cpu().numpy()
    return slabs_from_histogram(hist, world)


class ShardedMesh:
    """This rank's part of the assembled mesh: the vertices it OWNS (global ids vertex_offset .. vertex_offset + V_owned),
    their edge keys, and its triangles with GLOBAL vertex ids (global triangle ids triangle_offset ..)."""

    def __init__(self, vertices, keys, triangles, vertex_offset, triangle_offset, n_vertices_total, n_triangles_total):
        self.vertices, self.keys, self.triangles = vertices, keys, triangles
        self.vertex_offset, self.triangle_offset = vertex_offset, triangle_offset
        self.n_vertices_total, self.n_triangles_total = n_vertices_total, n_triangles_total


class ShardedStepResult:
    def __init__(self, local, shard, ids, n_total, timings, balance):
        self.local = local          # engine result of this rank (SurfaceReconstruction-like)
        self.shard = shard
        self.ids = ids              # global particle ids of the local particle set
        self.n_total = n_total
        self.timings = timings
        self.balance = balance      # per-rank owned / held particle counts and bricks (identical on every rank)

    @property
    def stats(self):
        return self.local.stats

    def subdomain_stats(self):
        return self.local.subdomain_stats()


class ShardedReconstruction:
    def __init__(self, engine, device, group=None):
        self.engine = engine
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local = None
        mode = os.environ.get("SPLASH_EXCHANGE", "p2p")
        if mode not in ("p2p", "allgather"):
            raise ValueError("SPLASH_EXCHANGE must be 'p2p' or 'allgather', not %r" % mode)
        self._exchange_mode = mode
        self.exchange_bytes = 0  # payload bytes this rank sent in the last step()

    def load_local_particles(self, pts):
        """The particles this rank contributes (its share of the input), (n,3) in the engine's Real type (float32 unless the
        engine was created for float64)."""
        dt = getattr(self.engine, "np_dtype", np.float32)
        t = torch.as_tensor(np.ascontiguousarray(pts, dtype=dt)) if not torch.is_tensor(pts) else pts
        self.local = t.to(self.device).contiguous()

    # ---- collectives ----
    # RCCL ("nccl") moves device tensors directly.  gloo (CPU tests, and the 2-processes-on-one-GPU test) only moves
    # host memory: device tensors are staged through the host for the collective and brought back.
    def _stage(self):
        return self.device.type == "cuda" and dist.is_initialized() and dist.get_backend(self.group) == "gloo"

    def _all_gather_small(self, t):
        """all-gather of a small fixed-shape tensor -> stacked (world, ...)"""
        if self.world == 1:
            return t.unsqueeze(0)
        if self._stage():
            th = t.cpu()
            out = [torch.zeros_like(th) for _ in range(self.world)]
            dist.all_gather(out, th, group=self.group)
            return torch.stack(out, dim=0).to(self.device)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.stack(out, dim=0)

    def _all_reduce(self, t, op):
        if self.world == 1:
            return t
        if self._stage():
            th = t.cpu()
            dist.all_reduce(th, op=op, group=self.group)
            t.copy_(th)
            return t
        dist.all_reduce(t, op=op, group=self.group)
        return t

    def _all_gather_rows(self, t):
        """Padded all-gather of (n_r, ...) tensors with different n_r; returns concatenation + counts."""
        if self.world == 1:
            return t, [int(t.shape[0])]
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=self.device)
        counts = [int(c) for c in self._all_gather_small(n).flatten().tolist()]
        m = max(max(counts), 1)
        stage = self._stage()
        dev = torch.device("cpu") if stage else self.device
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        pad[: t.shape[0]] = t.to(dev)
        out = [torch.zeros_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        parts = [out[r][: counts[r]] for r in range(self.world)]
        return torch.cat(parts, dim=0).contiguous().to(self.device), counts

    def _exchange(self, send):
        """Sparse all-to-all: send[q] (k_q, ...) goes to rank q; returns the list received from every rank.
        Transport: one batch of point-to-point isend/irecv (RCCL and gloo); SPLASH_EXCHANGE=allgather selects a padded
        all-gather of every rank's outgoing rows instead.  Errors propagate -- there is no silent fallback."""
        if self.world == 1:
            return [send[0]]
        self.exchange_bytes += sum(int(t.numel()) * t.element_size() for q, t in enumerate(send) if q != self.rank)
        counts = torch.tensor([int(t.shape[0]) for t in send], dtype=torch.int64, device=self.device)
        matrix = self._all_gather_small(counts)  # matrix[r][q] = rows rank r sends to rank q
        if self._exchange_mode == "p2p":
            return self._exchange_p2p(send, matrix)
        return self._exchange_allgather(send, matrix)

    def _exchange_p2p(self, send, matrix):
        recv_counts = [int(c) for c in matrix[:, self.rank].tolist()]
        stage = self._stage()
        dev = torch.device("cpu") if stage else self.device
        ops, recv = [], []
        for q in range(self.world):
            if q == self.rank:
                recv.append(send[q])
                continue
            buf = torch.empty((recv_counts[q],) + tuple(send[q].shape[1:]), dtype=send[q].dtype, device=dev)
            recv.append(buf)
            if send[q].shape[0] > 0:
                ops.append(dist.P2POp(dist.isend, send[q].to(dev).contiguous(), q, group=self.group))
            if recv_counts[q] > 0:
                ops.append(dist.P2POp(dist.irecv, buf, q, group=self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return [b.to(self.device) for b in recv] if stage else recv

    def _exchange_allgather(self, send, matrix):
        me = self.rank
        outgoing = torch.cat([send[q] for q in range(self.world) if q != me], dim=0) if self.world > 1 else send[0][:0]
        everything, _ = self._all_gather_rows(outgoing.contiguous())
        m = matrix.cpu().numpy()
        recv, base = [], 0
        for r in range(self.world):
            row_total = int(m[r].sum() - m[r][r])  # rank r's outgoing rows, ordered by destination (self skipped)
            if r == me:
                recv.append(send[me])
            else:
                off = base + int(sum(m[r][q] for q in range(me) if q != r))
                recv.append(everything[off:off + int(m[r][me])])
            base += row_total
        return recv

    def _tick(self, name):
        """Per-phase wall time of step() incl. device sync (SPLASH_PROFILE_SHARDED=1 or profile=True)."""
        if not self._profile:
            return
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (now - self._t_last) * 1e3
        self._t_last = now

    # rows = [id (int64 as two int32) | payload viewed as int32]: ids and payload travel in ONE message per neighbour
    @staticmethod
    def _pack(gid, payload, width):
        p = payload.reshape(gid.shape[0], width).contiguous().view(torch.int32)
        return torch.cat([gid.contiguous().view(torch.int32).reshape(-1, 2), p], dim=1).contiguous()

    @staticmethod
    def _unpack(rows, dtype, width):
        gid = rows[:, :2].contiguous().view(torch.int64).reshape(-1)
        pay = rows[:, 2:].contiguous().view(dtype).reshape(-1, width) if width > 1 else rows[:, 2:].contiguous().view(dtype).reshape(-1)
        return gid, pay

    def step(self, profile=None):
        """One sharded reconstruction (docstring of this module, steps 1-6)."""
        self._profile = bool(os.environ.get("SPLASH_PROFILE_SHARDED")) if profile is None else bool(profile)
        self.timings = getattr(self, "timings", {}) if self._profile else {}
        if self._profile and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._t_last = time.perf_counter()
        self.exchange_bytes = 0
        eng, dev, me, world = self.engine, self.device, self.rank, self.world
        local = self.local
        # 1. global particle ids = concatenation by rank (defines the summation order of the level set)
        n_loc = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        counts = [int(c) for c in self._all_gather_small(n_loc).flatten().tolist()]
        offset = sum(counts[:me])
        n_total = sum(counts)
        gid = torch.arange(offset, offset + local.shape[0], dtype=torch.int64, device=dev)
        # global particle AABB (identical on every rank)
        big = torch.finfo(local.dtype).max
        if local.shape[0]:
            # full reductions over the three strided columns (a dim-0 reduction of an (N,3) tensor runs
            # on 3 threads' worth of parallelism in torch and costs ~6 ms per call at 10M particles)
            mm = [torch.aminmax(local[:, d]) for d in range(3)]
            lo_hi = torch.stack([torch.stack([m.min for m in mm]), -torch.stack([m.max for m in mm])])
        else:
            lo_hi = torch.full((2, 3), big, dtype=local.dtype, device=dev)
        lo_hi = self._all_reduce(lo_hi, dist.ReduceOp.MIN)
        np_dt = np.float64 if local.dtype == torch.float64 else np.float32
        dmin = lo_hi[0].cpu().numpy() if n_total else np.zeros(3, np_dt)
        dmax = (-lo_hi[1]).cpu().numpy() if n_total else np.zeros(3, np_dt)
        gmin, sub_size, ns, margin, n_cubes = eng.grid_for_domain(dmin, dmax)
        self._grid = dict(ns=list(ns), n_cubes=int(n_cubes))
        self._tick("1_ids_aabb_grid")
        # 2. bricks balanced by owner counts (histogram all-reduced, so identical everywhere)
        sub = [torch.floor((local[:, d] - float(gmin[d])) / sub_size).to(torch.int64).clamp_(0, ns[d] - 1) for d in range(3)]
        flat = (sub[0] * ns[1] + sub[1]) * ns[2] + sub[2]
        hist = torch.bincount(flat, minlength=ns[0] * ns[1] * ns[2]).to(torch.int64)
        hist = self._all_reduce(hist, dist.ReduceOp.SUM)
        hist3 = hist.cpu().numpy().reshape(ns)
        # tie-break between equally good cuts: the axis along which the ranks' inputs are already separated (smallest
        # local / global extent, maximised over ranks), so that fewer particles move
        ext = torch.zeros(3, dtype=torch.float64, device=dev)
        if local.shape[0] and n_total:
            span = torch.tensor([max(float(dmax[d] - dmin[d]), 1e-300) for d in range(3)], dtype=torch.float64, device=dev)
            ext = torch.stack([(mm[d].max - mm[d].min).to(torch.float64) for d in range(3)]) / span
        ext = self._all_reduce(ext, dist.ReduceOp.MAX).cpu().numpy()
        bricks = bricks_from_histogram(hist3, world, axis_pref=tuple(round(float(e), 3) for e in ext))
        self._bricks = bricks
        lo, hi = bricks[me]
        shard = ShardDesc(dmin, dmax, lo, hi)
        # conservative coordinate box of a brick incl. ghost margin (the engine applies the exact rule)
        pad = margin * 1.001 + 1e-6 * max(1.0, float(np.abs(gmin).max()), float(np.abs(dmax).max()))

        def select(xyz, q):
            a, b = bricks[q]
            if any(b[d] <= a[d] for d in range(3)):
                return torch.zeros(xyz.shape[0], dtype=torch.bool, device=dev)
            m = None
            for d in range(3):
                c = xyz[:, d]
                md = (c >= float(gmin[d]) + a[d] * sub_size - pad) & (c <= float(gmin[d]) + b[d] * sub_size + pad)
                m = md if m is None else (m & md)
            return m

        self._tick("2_partition")
        # 3. (id, position) to every rank that needs the particle (owner or ghost), one message per destination
        width = 3
        send = []
        for q in range(world):
            m = select(local, q)
            send.append(self._pack(gid[m], local[m], width))
        recv = self._exchange(send)
        # Lists arrive ascending from every source rank and ranks' id ranges are ascending, so the
        # concatenation by source rank IS the ascending global-id order (no sort needed).
        rows = torch.cat(recv).contiguous()
        gids, L = self._unpack(rows, local.dtype, width)
        L = L.contiguous()
        self._tick("3_position_exchange")
        # 4. phase 1: densities of the particles contained in this brick (others stay 0)
        rho = eng.begin(L, shard)
        self._tick("4_phase1_binning_densities")
        owned = rho > 0
        # 5. halo densities: owners -> ranks holding the particle as a ghost
        send = []
        for q in range(world):
            if q == me:
                send.append(self._pack(gids[:0], rho[:0], 1))
                continue
            m = owned & select(L, q)
            send.append(self._pack(gids[m], rho[m], 1))
        recv = self._exchange(send)
        for q in range(world):
            if q == me or recv[q].shape[0] == 0:
                continue
            g_q, r_q = self._unpack(recv[q], rho.dtype, 1)
            pos = torch.searchsorted(gids, g_q)
            rho.index_copy_(0, pos, r_q)
        self._tick("5_density_exchange")
        # 6. phase 2
        res = eng.finish(rho)
        self._tick("6_phase2_levelset_mc")
        self.last = dict(gids=gids, rho=rho, owned=owned)
        # load balance of the partition (identical on every rank): owned / held particles per rank
        mine = torch.tensor([int(owned.sum().item()), int(gids.shape[0])], dtype=torch.int64, device=dev)
        per_rank = self._all_gather_small(mine).cpu().numpy()
        own = per_rank[:, 0].astype(np.float64)
        balance = dict(owned=[int(x) for x in per_rank[:, 0]], held=[int(x) for x in per_rank[:, 1]],
                       bricks=[[list(a), list(b)] for a, b in bricks],
                       imbalance_owned=float(own.max() / max(own.mean(), 1.0)),
                       imbalance_held=float(per_rank[:, 1].max() / max(per_rank[:, 1].mean(), 1.0)))
        self.last_balance = balance
        return ShardedStepResult(res, shard, gids, n_total, dict(self.timings), balance)

    # ---- result assembly ----
    def _local_mesh(self, step_result):
        if hasattr(self.engine, "device_mesh"):
            return self.engine.device_mesh(self.device)
        r = step_result.local
        v = torch.as_tensor(np.ascontiguousarray(r.mesh.vertices)).to(self.device)
        k = torch.as_tensor(np.ascontiguousarray(r.vertex_keys).astype(np.int64)).to(self.device)
        t = torch.as_tensor(np.ascontiguousarray(r.mesh.triangles).astype(np.int64)).to(self.device)
        return v, k, t

    def assemble(self, step_result):
        """Step 7 of the module docstring.  Returns this rank's `ShardedMesh`; runs on the device of this rank."""
        dev, me, world = self.device, self.rank, self.world
        v, k, t = self._local_mesh(step_result)
        ns, n = self._grid["ns"], self._grid["n_cubes"]
        npd = [ns[d] * n + 1 for d in range(3)]  # points per dimension of the global grid
        # key = ((gi*NPy + gj)*NPz + gk)*3 + axis  (include/splashsurf_hip.h: ss_result_vertex_keys)
        axis = k % 3
        p = k // 3
        g2 = p % npd[2]
        g1 = (p // npd[2]) % npd[1]
        g0 = p // (npd[2] * npd[1])
        g = [g0, g1, g2]

        def holds(q):  # does rank q's brick (closed box of grid points) contain both end points of the edge?
            a, b = self._bricks[q]
            if any(b[d] <= a[d] for d in range(3)):
                return torch.zeros(k.shape[0], dtype=torch.bool, device=dev)
            m = torch.ones(k.shape[0], dtype=torch.bool, device=dev)
            for d in range(3):
                m &= (g[d] >= a[d] * n) & (g[d] + (axis == d).to(torch.int64) <= b[d] * n)
            return m

        owner = torch.full((k.shape[0],), world, dtype=torch.int64, device=dev)
        holder = []
        for q in range(world - 1, -1, -1):
            hq = holds(q)
            holder.append(hq)
            owner = torch.where(hq, torch.full_like(owner, q), owner)
        holder = holder[::-1]
        if k.shape[0] and not bool(holder[me].all()):
            raise RuntimeError("rank %d emitted a vertex outside its brick" % me)
        mine = owner == me
        n_owned = int(mine.sum().item())
        cnt = torch.tensor([n_owned, int(t.shape[0])], dtype=torch.int64, device=dev)
        allc = self._all_gather_small(cnt).cpu().numpy()
        voff = int(allc[:me, 0].sum())
        toff = int(allc[:me, 1].sum())
        gid_local = torch.full((k.shape[0],), -1, dtype=torch.int64, device=dev)
        gid_local[mine] = voff + torch.arange(n_owned, dtype=torch.int64, device=dev)
        # owners -> the other ranks holding the edge: (key, global id)
        send = []
        for q in range(world):
            if q == me:
                send.append(torch.zeros((0, 2), dtype=torch.int64, device=dev))
                continue
            m = mine & holder[q]
            send.append(torch.stack([k[m], gid_local[m]], dim=1).contiguous())
        recv = self._exchange(send) if world > 1 else [send[0]]
        got = torch.cat([recv[q] for q in range(world) if q != me], dim=0) if world > 1 else send[0]
        need = ~mine
        if bool(need.any()):
            order = torch.argsort(got[:, 0])
            sk, sg = got[order, 0].contiguous(), got[order, 1]
            pos = torch.searchsorted(sk, k[need]).clamp_(max=max(int(sk.shape[0]) - 1, 0))
            if sk.shape[0] == 0 or not bool((sk[pos] == k[need]).all()):
                raise RuntimeError("rank %d: a shared face vertex was not emitted by its owner rank (level sets differ between ranks)" % me)
            gid_local[need] = sg[pos]
        tri = gid_local[t] if t.shape[0] else t
        return ShardedMesh(v[mine], k[mine], tri, voff, toff, int(allc[:, 0].sum()), int(allc[:, 1].sum()))

    def gather_densities(self):
        """Global density vector on every rank (tests): owned entries from every rank, ordered by global id."""
        g = self.last["gids"][self.last["owned"]]
        r = self.last["rho"][self.last["owned"]]
        G, _ = self._all_gather_rows(g)
        R, _ = self._all_gather_rows(r)
        out = torch.zeros(int(G.max().item()) + 1 if G.numel() else 0, dtype=r.dtype, device=self.device)
        out.index_copy_(0, G, R)
        return out

    def gather_mesh(self, step_result):
        """All ranks: the assembled mesh (vertices, keys, triangles) as numpy arrays on rank 0 (None elsewhere) -- the
        concatenation over ranks of what `assemble` left on each of them."""
        m = self.assemble(step_result)
        V, _ = self._all_gather_rows(m.vertices)
        K, _ = self._all_gather_rows(m.keys)
        T, _ = self._all_gather_rows(m.triangles)
        if self.rank != 0:
            return None
        assert V.shape[0] == m.n_vertices_total and T.shape[0] == m.n_triangles_total
        return V.cpu().numpy(), K.cpu().numpy().astype(np.uint64), T.cpu().numpy().astype(np.uint64)


# ---------------------------------------------------------------------------------------------------------------------
# the same algorithm inside the library: ss_dist_* (csrc/ss_dist.hip), RCCL behind the C ABI -- no Python on the data path
# ---------------------------------------------------------------------------------------------------------------------
class _DistInfo(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("brick_lo", C.c_int64 * 3), ("brick_hi", C.c_int64 * 3), ("n_total", C.c_uint64),
                ("n_held", C.c_uint64), ("n_owned", C.c_uint64), ("bytes_sent_positions", C.c_uint64), ("bytes_sent_densities", C.c_uint64),
                ("bytes_sent_assembly", C.c_uint64), ("ms_partition", C.c_double), ("ms_position_exchange", C.c_double), ("ms_density_exchange", C.c_double),
                ("ms_assembly", C.c_double), ("ms_phase1", C.c_double), ("ms_phase2", C.c_double), ("ms_own_turns", C.c_double), ("n_vertices_owned", C.c_uint64), ("vertex_offset", C.c_uint64), ("n_vertices_total", C.c_uint64),
                ("n_triangles", C.c_uint64), ("triangle_offset", C.c_uint64), ("n_triangles_total", C.c_uint64), ("n_collectives", C.c_uint64), ("ms_device", C.c_double), ("bytes_link_max", C.c_uint64)]


def _dist_lib(ctx):
    L = ctx._lib
    if not getattr(L, "_dist_configured", False):
        from . import api
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        L.ss_comm_unique_id.argtypes = [vp]
        L.ss_comm_create_rccl.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.ss_comm_adopt_rccl.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        L.ss_comm_create_local_group.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
        L.ss_comm_destroy.argtypes = [vp]
        L.ss_comm_destroy.restype = None
        L.ss_comm_local_group_take_turns.argtypes = [vp, i32]
        L.ss_comm_set_balance_feedback.argtypes = [vp, i32]
        L.ss_dist_reconstruct_f32.argtypes = [vp, vp, u64, C.POINTER(api._Params), vp]
        L.ss_dist_reconstruct_f64.argtypes = [vp, vp, u64, C.POINTER(api._Params64), vp]
        L.ss_dist_assemble.argtypes = [vp, vp]
        L.ss_dist_get_info.argtypes = [vp, C.POINTER(_DistInfo)]
        L.ss_dist_get_partition.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(u64), C.POINTER(u64)]
        for name in ("ss_dist_copy_global_ids", "ss_dist_copy_vertices", "ss_dist_copy_vertex_keys", "ss_dist_copy_triangles"):
            getattr(L, name).argtypes = [vp, vp]
        L._dist_configured = True
    return L


class NativeComm:
    """`ss_comm`: a communicator bound to one Context.  `NativeComm.rccl(ctx)`: one rank per process, RCCL; the unique id
    travels through torch.distributed's default group (any backend).  `NativeComm.local_group(ctxs)`: one communicator per
    context for host threads sharing a device (tests)."""

    def __init__(self, ctx, handle, rank, world, kind):
        self.ctx, self._h, self.rank, self.world, self.kind = ctx, handle, rank, world, kind

    @classmethod
    def rccl(cls, ctx, rank=None, world=None):
        L = _dist_lib(ctx)
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
        uid = (C.c_uint8 * 128)()
        failed = rank == 0 and L.ss_comm_unique_id(uid) != 0
        if world > 1:  # (rank 0 takes part in the broadcast even when it has no id: the other ranks must not wait for one that never comes)
            box = [None if failed else bytes(uid)]
            dist.broadcast_object_list(box, src=0)
            failed = box[0] is None
            if not failed:
                uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
        if failed:
            raise RuntimeError("ss_comm_unique_id failed on rank 0 (RCCL not loadable?)")
        h = C.c_void_p()
        st = L.ss_comm_create_rccl(ctx._h, uid, int(rank), int(world), C.byref(h))
        if st != 0:
            ctx._raise(st)
        return cls(ctx, h, rank, world, "rccl")

    @classmethod
    def local_group(cls, ctxs, take_turns=False):
        """`take_turns`: ss_comm_local_group_take_turns -- the ranks compute one at a time on the shared device, so that per-rank
        timers read what a rank takes on its own GPU (bench.py --pseudo-ranks)."""
        L = _dist_lib(ctxs[0])
        n = len(ctxs)
        hs = (C.c_void_p * n)(*[c._h for c in ctxs])
        out = (C.c_void_p * n)()
        st = L.ss_comm_create_local_group(hs, n, out)
        if st != 0:
            raise RuntimeError("ss_comm_create_local_group failed: %d" % st)
        if take_turns:
            st = L.ss_comm_local_group_take_turns(C.c_void_p(out[0]), 1)
            if st != 0:
                raise RuntimeError("ss_comm_local_group_take_turns failed: %d" % st)
        return [cls(ctxs[q], C.c_void_p(out[q]), q, n, "local") for q in range(n)]

    def set_balance_feedback(self, on=True):
        """ss_comm_set_balance_feedback: from the second step on the bricks balance the cost measured in the previous step (time series)."""
        st = self.ctx._lib.ss_comm_set_balance_feedback(self._h, 1 if on else 0)
        if st != 0:
            raise RuntimeError("ss_comm_set_balance_feedback failed: %d" % st)

    def destroy(self):
        if self._h:
            self.ctx._lib.ss_comm_destroy(self._h)
            self._h = None


def brick_owner_of(points, subdomain_grid, bricks):
    """Rank whose brick contains each particle's subdomain -- the owner rule of ss_dist.hip's histogram (floor of the coordinate in units of the subdomain
    edge, clamped into the grid).  Host-side helper for time series that keep their particles resident where they are owned: a rank that hands
    `ss_dist_reconstruct_*` the particles of its OWN brick exchanges halos only (the partition of the previous frame is `NativeSharded.partition()`,
    the subdomain grid `result.subdomain_grid`).  A particle this rule puts on the "wrong" side of a brick face because of rounding costs bytes, never
    correctness: the library decides ownership itself."""
    p = np.asarray(points)
    g = np.asarray(subdomain_grid.aabb.min, dtype=p.dtype)
    s = np.floor((p - g) / p.dtype.type(subdomain_grid.cell_size)).astype(np.int64)
    s = np.clip(s, 0, np.asarray(subdomain_grid.ncells_per_dim, dtype=np.int64) - 1)
    owner = np.full(p.shape[0], -1, dtype=np.int64)
    for q, (lo, hi) in enumerate(bricks):
        m = np.all(s >= np.asarray(lo), axis=1) & np.all(s < np.asarray(hi), axis=1)
        owner[m] = q
    return owner


class NativeSharded:
    """Per-rank driver of ss_dist_reconstruct_* / ss_dist_assemble.  `particles`: this rank's share, (n,3) numpy array or
    torch tensor (host or this rank's GPU) of float32 / float64."""

    def __init__(self, comm, params):
        from . import api
        self.api, self.comm, self.ctx, self.params = api, comm, comm.ctx, params
        self.lib = _dist_lib(comm.ctx)
        h = C.c_void_p()
        st = self.lib.ss_result_create(self.ctx._h, C.byref(h))
        if st != 0:
            self.ctx._raise(st)
        self.result = api.SurfaceReconstruction(self.ctx, h)
        self._keep = None

    def step(self, particles):
        ptr, n, keep, f64 = self.ctx._as_ptr(particles)
        self._keep, self.f64 = keep, f64
        p = self.params._c(f64)
        fn = self.lib.ss_dist_reconstruct_f64 if f64 else self.lib.ss_dist_reconstruct_f32
        st = fn(self.comm._h, ptr, n, C.byref(p), self.result._h)
        if st != 0:
            self.ctx._raise(st)
        self.result._invalidate()
        return self.result

    def assemble(self):
        st = self.lib.ss_dist_assemble(self.comm._h, self.result._h)
        if st != 0:
            self.ctx._raise(st)
        return self.info()

    def info(self):
        i = _DistInfo()
        self.lib.ss_dist_get_info(self.comm._h, C.byref(i))
        d = {k: getattr(i, k) for k, _ in _DistInfo._fields_ if not k.startswith("brick")}
        d["brick"] = [list(i.brick_lo), list(i.brick_hi)]
        return d

    def partition(self):
        w = self.comm.world
        b, o, h = (C.c_int64 * (6 * w))(), (C.c_uint64 * w)(), (C.c_uint64 * w)()
        st = self.lib.ss_dist_get_partition(self.comm._h, b, o, h)
        if st != 0:
            raise RuntimeError("ss_dist_get_partition failed")
        own = np.array(list(o), dtype=np.float64)
        held = np.array(list(h), dtype=np.float64)
        return dict(bricks=[[list(b[6 * q:6 * q + 3]), list(b[6 * q + 3:6 * q + 6])] for q in range(w)], owned=[int(x) for x in o], held=[int(x) for x in h],
                    imbalance_owned=float(own.max() / max(own.mean(), 1.0)), imbalance_held=float(held.max() / max(held.mean(), 1.0)))

    def _copy(self, fn, count, dtype):
        out = np.empty(count, dtype=dtype)
        if out.size:
            st = fn(self.comm._h, out.ctypes.data_as(C.c_void_p))
            if st != 0:
                self.ctx._raise(st)
        return out

    def global_ids(self):
        return self._copy(self.lib.ss_dist_copy_global_ids, self.info()["n_held"], np.uint64)

    def mesh_piece(self):
        """(owned vertices, their edge keys, triangles with global vertex ids) of this rank, as numpy arrays."""
        i = self.info()
        dt = np.float64 if self.f64 else np.float32
        v = self._copy(self.lib.ss_dist_copy_vertices, i["n_vertices_owned"] * 3, dt).reshape(-1, 3)
        k = self._copy(self.lib.ss_dist_copy_vertex_keys, i["n_vertices_owned"], np.uint64)
        t = self._copy(self.lib.ss_dist_copy_triangles, i["n_triangles"] * 3, np.uint64).reshape(-1, 3)
        return v, k, t

    def close(self):
        self.result._free()
        self.comm.destroy()
