"""Multi-GPU surface reconstruction: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

The reference is single-process; its only natural decomposition is the uniform grid of subdomains
(SURVEY.md section 8e).  Here the subdomain grid of ONE global domain is cut into contiguous slabs
along its longest axis, balanced by particle count:

  1. global particle ids = concatenation of the ranks' inputs (the summation order of the level set is
     "ascending particle index", so ids must be global); global AABB by all-reduce(MIN/MAX);
  2. every rank derives the same global grid; a histogram of owner subdomains along the axis is
     all-reduced and cut into `world` slabs of (nearly) equal particle count;
  3. sparse all-to-all (batched isend/irecv) of (id, position): each particle goes to every rank whose
     slab + ghost margin contains it -- in a weak-scaling run only thin halo layers travel;
  4. phase 1 on each rank: binning + densities of the particles CONTAINED in its slab;
  5. halo density exchange: owners send (id, rho) to the ranks that hold the particle as a ghost
     (the values are copied, never re-computed, so densities are bit-identical to a single-process run);
  6. phase 2: level set + marching cubes for the slab.  Vertices on slab faces are produced by both
     neighbours with identical global edge keys and identical coordinates; `gather_mesh` removes the
     duplicates by key.

The per-rank engine is pluggable: `HipEngine` drives the C ABI (ss_shard_begin_f32 / ss_shard_finish),
the CPU tests plug in the oracle (tests/test_distributed.py).
"""
import ctypes as C

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
        if local_pts.is_cuda:
            torch.cuda.current_stream(local_pts.device).synchronize()
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
        if rho.is_cuda:
            torch.cuda.current_stream(rho.device).synchronize()
        fn = self.lib.ss_shard_set_densities_f64 if self.f64 else self.lib.ss_shard_set_densities
        st = fn(self.result._h, C.c_void_p(rho.data_ptr()), int(rho.shape[0]))
        if st != 0:
            self.ctx._raise(st)
        st = self.lib.ss_shard_finish(self.ctx._h, self.result._h)
        if st != 0:
            self.ctx._raise(st)
        self.result._invalidate()
        return self.result


def partition_slabs(coords_axis, gmin_axis, sub_size, ns_axis, world):
    """Contiguous slabs of subdomain indices along one axis, balanced by owner-particle counts.
    Deterministic given identical inputs on all ranks. Returns list of (lo, hi)."""
    s = torch.floor((coords_axis - gmin_axis) / sub_size).to(torch.int64).clamp_(0, ns_axis - 1)
    hist = torch.bincount(s, minlength=ns_axis).cpu().numpy()
    return slabs_from_histogram(hist, world)


def slabs_from_histogram(hist, world):
    ns_axis = int(len(hist))
    cum = np.concatenate([[0.0], np.cumsum(np.asarray(hist, dtype=np.float64))])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        # choose the boundary (k-1 or k) closest to the target, keep bounds monotone and leave room for the rest
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, ns_axis)] - target):
            k -= 1
        k = max(k, bounds[-1])
        k = min(k, ns_axis)
        bounds.append(k)
    bounds.append(ns_axis)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


class ShardedStepResult:
    def __init__(self, local, shard, ids, n_total, timings):
        self.local = local          # engine result of this rank (SurfaceReconstruction-like)
        self.shard = shard
        self.ids = ids              # global particle ids of the local particle set
        self.n_total = n_total
        self.timings = timings

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
        import os
        self._exchange_mode = "allgather" if os.environ.get("SPLASH_EXCHANGE", "p2p") == "allgather" else "p2p"

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
        Default transport: one batch of point-to-point isend/irecv (RCCL and gloo).  SPLASH_EXCHANGE=allgather (or a
        failing point-to-point batch) switches to a padded all-gather of every rank's outgoing rows, from which each
        rank keeps its part -- more bytes on the wire, but only the most basic collective."""
        if self.world == 1:
            return [send[0]]
        counts = torch.tensor([int(t.shape[0]) for t in send], dtype=torch.int64, device=self.device)
        matrix = self._all_gather_small(counts)  # matrix[r][q] = rows rank r sends to rank q
        if self._exchange_mode == "p2p":
            try:
                return self._exchange_p2p(send, matrix)
            except RuntimeError as e:  # symmetric failures (unsupported transport): every rank falls back
                import warnings
                warnings.warn("point-to-point halo exchange failed (%s); falling back to all-gather" % (str(e).splitlines()[0],))
                self._exchange_mode = "allgather"
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
        """Optional per-phase timing of step() (SPLASH_PROFILE_SHARDED=1): wall time incl. device sync."""
        if not self._profile:
            return
        import time
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        now = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (now - self._t_last) * 1e3
        self._t_last = now

    def step(self):
        """One sharded reconstruction.  Only halo layers travel between ranks:
        positions to the ranks whose slab (+ ghost margin) contains them, then the densities of owned
        particles to the ranks that hold them as ghosts."""
        import os
        import time
        self._profile = bool(os.environ.get("SPLASH_PROFILE_SHARDED"))
        self.timings = getattr(self, "timings", {}) if self._profile else {}
        if self._profile and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._t_last = time.perf_counter()
        eng, dev, me = self.engine, self.device, self.rank
        local = self.local
        # 1. global particle ids = concatenation by rank (defines the summation order of the level set)
        n_loc = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        counts = [int(c) for c in self._all_gather_small(n_loc).flatten().tolist()]
        offset = sum(counts[:me])
        n_total = sum(counts)
        gid = torch.arange(offset, offset + local.shape[0], dtype=torch.int64, device=dev)
        # 2. global particle AABB (identical on every rank)
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
        # slab axis: the longest axis of the subdomain grid; among equally long axes the one along which the ranks'
        # inputs are already separated (smallest local/global extent, maximised over ranks), so that fewer particles move
        ext = torch.zeros(3, dtype=torch.float64, device=dev)
        if local.shape[0] and n_total:
            span = torch.tensor([max(float(dmax[d] - dmin[d]), 1e-300) for d in range(3)], dtype=torch.float64, device=dev)
            ext = torch.stack([(mm[d].max - mm[d].min).to(torch.float64) for d in range(3)]) / span
        ext = self._all_reduce(ext, dist.ReduceOp.MAX).cpu().numpy()
        longest = max(ns)
        axis = min((d for d in range(3) if ns[d] == longest), key=lambda d: (ext[d], d))
        self._tick("1_ids_aabb_grid")
        # 3. slab partition balanced by owner counts (histogram all-reduced, so identical everywhere)
        s_own = torch.floor((local[:, axis] - float(gmin[axis])) / sub_size).to(torch.int64).clamp_(0, ns[axis] - 1)
        hist = torch.bincount(s_own, minlength=ns[axis]).to(torch.int64)
        hist = self._all_reduce(hist, dist.ReduceOp.SUM)
        slabs = slabs_from_histogram(hist.cpu().numpy(), self.world)
        lo, hi = slabs[me]
        sub_lo, sub_hi = [0, 0, 0], list(ns)
        sub_lo[axis], sub_hi[axis] = lo, hi
        shard = ShardDesc(dmin, dmax, sub_lo, sub_hi)
        # conservative coordinate interval of a slab incl. ghost margin (the engine applies the exact rule)
        pad = margin * 1.001 + 1e-6 * max(1.0, float(np.abs(gmin).max()), float(np.abs(dmax).max()))

        def interval(q):
            a, b = slabs[q]
            if b <= a:
                return None
            return float(gmin[axis]) + a * sub_size - pad, float(gmin[axis]) + b * sub_size + pad

        def select(coords, q):
            iv = interval(q)
            if iv is None:
                return torch.zeros(coords.shape[0], dtype=torch.bool, device=dev)
            return (coords >= iv[0]) & (coords <= iv[1])

        self._tick("2_partition")
        # 4. positions to every rank that needs them (owner or ghost)
        send_gid, send_xyz = [], []
        for q in range(self.world):
            m = select(local[:, axis], q)
            send_gid.append(gid[m])
            send_xyz.append(local[m])
        recv_gid = self._exchange(send_gid)
        recv_xyz = self._exchange(send_xyz)
        # Lists arrive ascending from every source rank and ranks' id ranges are ascending, so the
        # concatenation by source rank IS the ascending global-id order (no sort needed).
        gids = torch.cat(recv_gid).contiguous()
        L = torch.cat(recv_xyz).contiguous()
        self._tick("3_position_exchange")
        # 5. phase 1: densities of the particles contained in this slab (others stay 0)
        rho = eng.begin(L, shard)
        self._tick("4_phase1_binning_densities")
        owned = rho > 0
        # 6. halo densities: owners -> ranks holding the particle as a ghost
        send_gid, send_rho = [], []
        for q in range(self.world):
            if q == me:
                send_gid.append(gids[:0])
                send_rho.append(rho[:0])
                continue
            m = owned & select(L[:, axis], q)
            send_gid.append(gids[m])
            send_rho.append(rho[m])
        recv_gid = self._exchange(send_gid)
        recv_rho = self._exchange(send_rho)
        for q in range(self.world):
            if q == me or recv_gid[q].shape[0] == 0:
                continue
            pos = torch.searchsorted(gids, recv_gid[q])
            rho.index_copy_(0, pos, recv_rho[q])
        self._tick("5_density_exchange")
        # 7. phase 2
        res = eng.finish(rho)
        self._tick("6_phase2_levelset_mc")
        self.last = dict(gids=gids, rho=rho, owned=owned)
        return ShardedStepResult(res, shard, gids, n_total, dict(self.timings))

    # ---- result assembly (tests / consumers that want one mesh) ----
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
        """All ranks: returns (vertices, keys, triangles) of the merged mesh on rank 0 (None elsewhere).
        Duplicated face vertices are removed by global edge key; the lowest rank's copy is kept."""
        r = step_result.local
        v = torch.as_tensor(np.ascontiguousarray(r.mesh.vertices, dtype=np.float32)).to(self.device)
        k = torch.as_tensor(np.ascontiguousarray(r.vertex_keys).astype(np.int64)).to(self.device)
        t = torch.as_tensor(np.ascontiguousarray(r.mesh.triangles).astype(np.int64)).to(self.device)
        V, vc = self._all_gather_rows(v)
        K, _ = self._all_gather_rows(k)
        T, tc = self._all_gather_rows(t)
        if self.rank != 0:
            return None
        V, K, T = V.cpu().numpy(), K.cpu().numpy(), T.cpu().numpy()
        voff = np.concatenate([[0], np.cumsum(vc)])
        toff = np.concatenate([[0], np.cumsum(tc)])
        for q in range(len(vc)):
            T[toff[q]:toff[q + 1]] += voff[q]
        uk, first = np.unique(K, return_index=True)  # first occurrence = lowest rank
        remap = np.searchsorted(uk, K)
        return V[first], uk.astype(np.uint64), remap[T].astype(np.uint64)
