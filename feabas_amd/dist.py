"""Sharding and the (only) exchange steps of the multi-GPU runs.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  Tile pairs / sections are independent units: every rank works on a contiguous
shard -- the same contiguous-slice partitioning the reference uses for its worker jobs
(feabas/stitcher.py:375-392) -- with no collective on the data path.  What is exchanged afterwards:
  * the variable-length match table of every rank   -> ``gather_match_table``
  * per-section node displacement vectors           -> ``allgather_ragged``
"""
import os

import numpy as np


def shard_range(n_items, rank, world):
    """[start, stop) of the contiguous shard of `rank`; shard sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def _device(dist):
    import torch
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


class Exchange:
    """The exchange steps of a sharded run: ``gatherv`` (every rank's variable-length table -> the
    root: counts first, then point-to-point transfers of exactly the bytes each rank holds -- nothing is padded and only
    the root receives) and ``allgather`` (equal contributions: node displacements).

    On the GPU box the transfers go through the C ABI (``fb_comm_*`` / ``fb_gatherv_dev`` / ``fb_allgather_dev``: one RCCL
    communicator on the context's stream, device buffers; the 128-byte communicator id travels through the
    ``torch.distributed`` group that launched the ranks).  With a gloo group (CPU tests, no GPU) the same steps run as
    ``torch.distributed`` point-to-point operations on host tensors."""

    def __init__(self, group=None, backend=None, ctx=None):
        """collective over `group`.  ctx: the fb_ctx whose stream carries the transfers (default: the caller's current
        context); a host thread that only communicates should own one, so that a gather does not queue behind kernels"""
        torch, dist = _dist()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if backend is None:
            backend = 'rccl' if dist.get_backend(group) == 'nccl' else 'torch'
        self.backend = backend
        self.comm = None
        self.seconds = 0.0            # wall time spent inside gatherv / allgather (host view, includes the copies)
        self.bytes = 0
        if backend == 'rccl':
            import ctypes as C
            from . import _lib
            self._lib, self._C = _lib, C
            lib, ctx = _lib.load(), (_lib.ctx() if ctx is None else ctx)
            # every rank makes an id (the cheapest call that needs the RCCL library): a rank that cannot load it must not
            # leave the others waiting in the broadcast below, so the ranks agree on that first
            ident = np.zeros(128, dtype=np.uint8)
            rc = lib.fb_comm_unique_id(ctx, _lib.ptr(ident))
            err = None if rc == 0 else (lib.fb_last_error(ctx) or b'?').decode()
            flags = [None] * self.world
            dist.all_gather_object(flags, err, group=group)
            bad = [(r, e) for r, e in enumerate(flags) if e is not None]
            if bad:
                raise RuntimeError(f'RCCL is not available on rank {bad[0][0]}: {bad[0][1]}')
            box = [ident.tobytes()]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = np.frombuffer(box[0], dtype=np.uint8).copy()
            h = C.c_void_p()
            # ncclCommInitRank blocks until every rank has joined; a rank stuck in it is given up after a while (the thread
            # is left behind) so that the caller can fall back to another route instead of hanging the job
            import threading
            res = {}

            def init():
                res['rc'] = lib.fb_comm_create(ctx, _lib.ptr(ident), self.rank, self.world, C.byref(h))
                res['err'] = (lib.fb_last_error(ctx) or b'').decode() if res['rc'] else ''
            th = threading.Thread(target=init, daemon=True)
            th.start()
            th.join(float(os.environ.get('FEABAS_HIP_COMM_TIMEOUT', '120')))
            if th.is_alive():
                raise RuntimeError('fb_comm_create did not return (ncclCommInitRank waiting for the other ranks)')
            if res['rc'] != 0:
                raise _lib.FeabasHipError(res['rc'], res['err'] or '?')
            self.comm, self._ctx = h, ctx

    def close(self):
        if self.comm is not None:
            self._lib.load().fb_comm_destroy(self._ctx, self.comm)
            self.comm = None

    # -- counts of every rank (one int64 each)
    def _counts(self, n):
        torch, dist = _dist()
        if self.backend == 'rccl':
            _lib, lib = self._lib, self._lib.load()
            mine = _lib.DeviceBuffer.from_array(np.array([n], dtype=np.int64))
            allc = _lib.DeviceBuffer(8 * self.world)
            _lib.check(lib.fb_allgather_dev(self._ctx, self.comm, mine.ptr, allc.ptr, 8), h=self._ctx)
            out = allc.to_array((self.world,), np.int64)
            mine.free(); allc.free()
            return out
        dev = _device(dist)
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        cnts = torch.zeros(self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(cnts, cnt, group=self.group)
        return cnts.cpu().numpy()

    def _scope(self):
        import contextlib
        return self._lib.using(self._ctx) if self.backend == 'rccl' else contextlib.nullcontext()

    def gatherv(self, arr, root=0):
        with self._scope():
            return self._gatherv(arr, root)

    def allgather(self, arr):
        with self._scope():
            return self._allgather(arr)

    def _gatherv(self, arr, root=0):
        """arr [n_r, ...] (first dimension differs between ranks) -> on `root` the list of every rank's array, elsewhere
        None.  One count exchange + one grouped set of transfers to the root."""
        import time
        t0 = time.perf_counter()
        torch, dist = _dist()
        arr = np.ascontiguousarray(arr)
        tail, row = arr.shape[1:], int(np.prod(arr.shape[1:], dtype=np.int64)) * arr.dtype.itemsize
        cnts = self._counts(arr.shape[0])
        nbytes = cnts * row
        total = int(nbytes.sum())
        out = None
        if self.backend == 'rccl':
            _lib, lib = self._lib, self._lib.load()
            send = _lib.DeviceBuffer.from_array(arr) if arr.nbytes else None
            recv = _lib.DeviceBuffer(total) if self.rank == root and total else None
            cb = np.ascontiguousarray(nbytes, dtype=np.int64)
            _lib.check(lib.fb_gatherv_dev(self._ctx, self.comm, send.ptr if send else None, _lib.ptr(cb), recv.ptr if recv else None, root), h=self._ctx)
            if self.rank == root:
                flat = recv.to_array((total,), np.uint8) if recv else np.empty(0, np.uint8)
                offs = np.concatenate(([0], np.cumsum(nbytes)))
                out = [flat[offs[r]:offs[r + 1]].view(arr.dtype).reshape((int(cnts[r]),) + tail) for r in range(self.world)]
            else:
                _lib.check(lib.fb_sync(self._ctx), h=self._ctx)
            for b in (send, recv):
                if b is not None:
                    b.free()
        else:
            dev = _device(dist)
            ops, bufs = [], []
            if self.rank == root:
                for r in range(self.world):
                    if r == root or cnts[r] == 0:
                        continue
                    t = torch.empty(int(nbytes[r]), dtype=torch.uint8, device=dev)
                    bufs.append((r, t))
                    ops.append(dist.P2POp(dist.irecv, t, dist.get_global_rank(self.group, r) if self.group is not None else r, group=self.group))
            elif arr.nbytes:
                t = torch.from_numpy(arr.reshape(-1).view(np.uint8).copy()).to(dev)
                ops.append(dist.P2POp(dist.isend, t, dist.get_global_rank(self.group, root) if self.group is not None else root, group=self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.rank == root:
                got = {r: t.cpu().numpy() for r, t in bufs}
                out = []
                for r in range(self.world):
                    if r == root:
                        out.append(arr)
                    elif cnts[r] == 0:
                        out.append(np.empty((0,) + tail, dtype=arr.dtype))
                    else:
                        out.append(got[r].view(arr.dtype).reshape((int(cnts[r]),) + tail))
        self.seconds += time.perf_counter() - t0
        self.bytes += total if self.rank == root else arr.nbytes
        return out

    def _allgather(self, arr):
        """equal-shaped arr of every rank -> array [world, ...] on every rank (one collective)"""
        import time
        t0 = time.perf_counter()
        torch, dist = _dist()
        arr = np.ascontiguousarray(arr)
        if self.backend == 'rccl':
            _lib, lib = self._lib, self._lib.load()
            send = _lib.DeviceBuffer.from_array(arr)
            recv = _lib.DeviceBuffer(arr.nbytes * self.world)
            _lib.check(lib.fb_allgather_dev(self._ctx, self.comm, send.ptr, recv.ptr, arr.nbytes), h=self._ctx)
            out = recv.to_array((self.world,) + arr.shape, arr.dtype)
            send.free(); recv.free()
        else:
            dev = _device(dist)
            t = torch.from_numpy(arr.reshape(-1).view(np.uint8).copy()).to(dev)
            o = torch.empty(t.numel() * self.world, dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(o, t, group=self.group)
            out = o.cpu().numpy().view(arr.dtype).reshape((self.world,) + arr.shape)
        self.seconds += time.perf_counter() - t0
        self.bytes += arr.nbytes * self.world
        return out


_default_exchange = {}


def exchange(group=None):
    """the process's Exchange for `group` (created on first use: collective -- every rank of the group must call it)"""
    key = id(group)
    if key not in _default_exchange:
        _default_exchange[key] = Exchange(group)
    return _default_exchange[key]


def release_exchanges():
    for ex in _default_exchange.values():
        ex.close()
    _default_exchange.clear()


def allgather_ragged(arr, group=None):
    """all-gather of per-rank arrays whose first dimension differs: the list of every rank's array, on every rank.
    A gatherv to rank 0 would do for the match table (north star: one gather); this form is for callers that need the
    table everywhere (the coupled-window set-up, tests).  Counts first, then every rank's exact bytes (no padding)."""
    torch, dist = _dist()
    ex = exchange(group)
    arr = np.ascontiguousarray(arr)
    parts = None
    for root in range(ex.world):
        got = ex.gatherv(arr, root=root)
        if got is not None:
            parts = got
    return parts


def gather_match_table(pair_ids, xy0, xy1, weight, pair_offset=0, group=None, root=0, dtype=np.float64):
    """Match tables of all ranks as one table [M, 6] = (pair id, x0, y0, x1, y1, weight) on `root` (None on the other
    ranks; root=None: on every rank), ordered by rank (= by global pair id for contiguous shards).  dtype float64 keeps
    the matcher's precision; float32 is the wire format -- the reference stores concat(xy0, xy1, weight) per pair as
    float32 (stitcher.py:144-151), 24 bytes per row, pair ids exact up to 2^24."""
    tab = np.concatenate((np.asarray(pair_ids, dtype=dtype).reshape(-1, 1) + dtype(pair_offset),
                          np.asarray(xy0, dtype=dtype).reshape(-1, 2), np.asarray(xy1, dtype=dtype).reshape(-1, 2),
                          np.asarray(weight, dtype=dtype).reshape(-1, 1)), axis=1)
    if root is None:
        return np.concatenate(allgather_ragged(tab, group=group), axis=0)
    parts = exchange(group).gatherv(tab, root=root)
    return None if parts is None else np.concatenate(parts, axis=0)


# ---------------------------------------------------------------------------------------------------------------------
# Coupled-window solve (SURVEY.md sec.8e mode 2 / sec.8f row 4): the sections of an alignment window form ONE system
# (aligner.py:510-535, 696-727) whose rows are partitioned by section over the ranks.  Links only join neighbouring
# sections, so a rank needs the entries of a few other ranks' vectors: a point-to-point halo exchange per iteration and
# ONE fused all-reduce of three scalars (Chronopoulos-Gear form of the preconditioned conjugate gradients: the two inner
# products and the residual norm of an iteration are taken on the same vectors).
class RowPartition:
    """Who owns which rows, which foreign columns the local rows touch (the halo), and the send / receive lists.

    indptr / indices: CSR pattern of the LOCAL rows with GLOBAL column ids; row_start: global id of the first local row.
    After construction: ``halo`` (sorted global ids of foreign columns), ``local_cols`` (indices with every column mapped
    to [0, n_loc) for own columns and n_loc + position in halo for foreign ones), ``recv`` {rank: slice of halo},
    ``send`` {rank: local row ids that rank needs}."""

    def __init__(self, indptr, indices, row_start, group=None):
        torch, dist = _dist()
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        indptr = np.asarray(indptr, dtype=np.int64); indices = np.asarray(indices, dtype=np.int64)
        self.n_loc = indptr.size - 1
        self.row_start = int(row_start)
        spans = [None] * self.world
        dist.all_gather_object(spans, (self.row_start, self.n_loc), group=group)
        self.starts = np.array([s for s, _ in spans], dtype=np.int64)
        self.counts = np.array([c for _, c in spans], dtype=np.int64)
        if np.any(self.starts[1:] != self.starts[:-1] + self.counts[:-1]):
            raise ValueError('RowPartition: the row blocks of the ranks must be contiguous and in rank order')
        own = (indices >= self.row_start) & (indices < self.row_start + self.n_loc)
        self.halo = np.unique(indices[~own])
        owner = np.searchsorted(self.starts, self.halo, side='right') - 1
        self.local_cols = np.where(own, indices - self.row_start, self.n_loc + np.searchsorted(self.halo, indices)).astype(np.int64)
        self.recv = {}
        need = {}
        for r in np.unique(owner):
            sel = np.flatnonzero(owner == r)
            self.recv[int(r)] = slice(int(sel[0]), int(sel[-1]) + 1)         # halo is sorted, owners are contiguous runs
            need[int(r)] = self.halo[sel] - self.starts[r]
        wants = [None] * self.world
        dist.all_gather_object(wants, need, group=group)
        self.send = {src: np.asarray(w[self.rank], dtype=np.int64) for src, w in enumerate(wants) if self.rank in w and src != self.rank}
        self.n_halo = int(self.halo.size)

    def exchange(self, v_loc, out_halo):
        """fill out_halo (torch tensor [n_halo]) with the foreign entries of the distributed vector whose local part is v_loc"""
        torch, dist = _dist()
        if self.world == 1 or (not self.send and not self.recv):
            return
        dev = _device(dist)
        ops, bufs = [], []
        for dst, rows in sorted(self.send.items()):
            t = v_loc[torch.as_tensor(rows, device=v_loc.device)].to(dev).contiguous()
            ops.append(dist.P2POp(dist.isend, t, dst, group=self.group))
        for src, sl in sorted(self.recv.items()):
            t = torch.empty(sl.stop - sl.start, dtype=v_loc.dtype, device=dev)
            bufs.append((sl, t))
            ops.append(dist.P2POp(dist.irecv, t, src, group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for sl, t in bufs:
            out_halo[sl] = t.to(out_halo.device)


def pcg_row_partitioned(part, spmv, b_loc, minv_loc, rtol=1e-7, maxiter=10000, check_every=8, fused=None):
    """Jacobi-preconditioned conjugate gradients on a row-partitioned system.

    part: RowPartition; spmv(u_ext) -> (A u)_loc for u_ext = [u_loc | u_halo] (torch tensor, n_loc + n_halo);
    b_loc, minv_loc: torch tensors [n_loc] on the compute device (right-hand side, inverse diagonal).
    One halo exchange + one SpMV + ONE all-reduce (3 scalars) per iteration; the scalars stay on the device, the host
    looks at the residual every `check_every` iterations only (with RCCL nothing else synchronises the host).
    fused (default: when `spmv` is a DeviceRows and the vectors live on the GPU): the vector updates, the three inner
    products and the scalar recurrences run as the library's fused kernels (fb_cgcg_*_dev) on the rows' stream.
    A non-positive denominator of the recurrence (a system that is not positive definite in floating point, or a residual
    that has reached exactly zero) drops the step instead of dividing by zero; a non-finite residual ends the loop with the
    last iterate and raises.  Stops at ||r|| <= rtol ||b||.  Returns (x_loc, iterations, relative residual)."""
    torch, dist = _dist()
    if fused is None:
        fused = isinstance(spmv, DeviceRows) and b_loc.is_cuda
    if fused:
        return _pcg_row_partitioned_fused(part, spmv, b_loc, minv_loc, rtol, maxiter, check_every)
    cdev = _device(dist)
    n = part.n_loc

    def reduce3(a, b_, c):
        t = torch.stack((a, b_, c))
        if part.world > 1:
            if t.device != cdev:
                tc = t.to(cdev)
                dist.all_reduce(tc, group=part.group)
                t = tc.to(t.device)
            else:
                dist.all_reduce(t, group=part.group)
        return t

    ext = torch.zeros(n + part.n_halo, dtype=b_loc.dtype, device=b_loc.device)
    x = torch.zeros_like(b_loc)
    r = b_loc.clone()
    u = minv_loc * r
    ext[:n] = u; part.exchange(u, ext[n:])
    w = spmv(ext).clone()
    t = reduce3(torch.dot(r, u), torch.dot(w, u), torch.dot(r, r))
    gamma, delta, rr = t[0], t[1], t[2]
    bnorm2 = float(rr)
    if bnorm2 == 0.0 or maxiter == 0:
        return x, 0, 0.0
    zero = torch.zeros((), dtype=b_loc.dtype, device=b_loc.device)
    p = torch.zeros_like(b_loc); s = torch.zeros_like(b_loc)
    alpha = torch.where(delta > 0, gamma / delta, zero)
    beta = zero.clone()
    it = 0
    rel = 1.0
    while it < maxiter:
        p = u + beta * p
        s = w + beta * s
        x += alpha * p
        r -= alpha * s
        u = minv_loc * r
        ext[:n] = u; part.exchange(u, ext[n:])
        w = spmv(ext).clone()
        t = reduce3(torch.dot(r, u), torch.dot(w, u), torch.dot(r, r))
        it += 1
        if it % check_every == 0 or it == maxiter:
            rel = (float(t[2]) / bnorm2) ** 0.5                      # the only host synchronisation of the loop
            if not np.isfinite(rel):
                raise FloatingPointError('pcg_row_partitioned: the residual is not finite (system not positive definite?)')
            if rel <= rtol:
                break
        beta = torch.where(gamma != 0, t[0] / gamma, zero)
        den = t[1] - beta * t[0] / torch.where(alpha != 0, alpha, torch.ones_like(alpha))
        ok = (den > 0) & (alpha != 0)
        alpha = torch.where(ok, t[0] / torch.where(ok, den, torch.ones_like(den)), zero)
        beta = torch.where(ok, beta, zero)
        gamma = t[0]
    return x, it, rel


def _pcg_row_partitioned_fused(part, rows, b_loc, minv_loc, rtol, maxiter, check_every):
    """the same iteration with the library's fused kernels: per iteration  fb_cgcg_update_dev -> halo exchange ->
    fb_spmv_dev -> fb_cgcg_dots_dev -> all-reduce (3 doubles) -> fb_cgcg_scalars_dev, on the stream of `rows`"""
    torch, dist = _dist()
    from . import _lib
    lib, ctx = _lib.load(), rows._ctx
    cdev = _device(dist)
    n = part.n_loc
    dev = b_loc.device
    f64 = torch.float64
    with torch.cuda.stream(rows.stream()):
        ext = torch.zeros(n + part.n_halo, dtype=f64, device=dev)
        x = torch.zeros(n, dtype=f64, device=dev)
        r = b_loc.to(f64).clone()
        p = torch.zeros(n, dtype=f64, device=dev); s = torch.zeros(n, dtype=f64, device=dev)
        w = torch.zeros(n, dtype=f64, device=dev)
        minv = minv_loc.to(f64).contiguous()
        state = torch.zeros(8, dtype=f64, device=dev)
        t3 = torch.zeros(3, dtype=f64, device=dev)
        scratch = torch.zeros(3 * 1024, dtype=f64, device=dev)
        u = ext[:n]                                                  # u lives at the head of the extended vector: no copy before the SpMV

        def product_and_dots(first):
            part.exchange(u, ext[n:])
            y = rows(ext)                                            # fb_spmv_dev on the same stream
            w.copy_(y)
            _lib.check(lib.fb_cgcg_dots_dev(ctx, n, C_ptr(r), C_ptr(u), C_ptr(w), C_ptr(scratch), C_ptr(t3)), h=ctx)
            if part.world > 1:
                if cdev != dev:
                    tc = t3.to(cdev)
                    dist.all_reduce(tc, group=part.group)
                    t3.copy_(tc)
                else:
                    dist.all_reduce(t3, group=part.group)
            _lib.check(lib.fb_cgcg_scalars_dev(ctx, C_ptr(t3), C_ptr(state), 1 if first else 0), h=ctx)
        u.copy_(minv * r)
        product_and_dots(True)
        bnorm2 = float(state[3])
        if bnorm2 == 0.0 or maxiter == 0:
            return x, 0, 0.0
        it = 0
        rel = 1.0
        while it < maxiter:
            _lib.check(lib.fb_cgcg_update_dev(ctx, n, C_ptr(state), C_ptr(minv), C_ptr(x), C_ptr(r), C_ptr(u), C_ptr(w), C_ptr(p), C_ptr(s)), h=ctx)
            product_and_dots(False)
            it += 1
            if it % check_every == 0 or it == maxiter:
                st = state.cpu()                                     # the only host synchronisation of the loop
                rel = (float(st[3]) / bnorm2) ** 0.5
                if not np.isfinite(rel):
                    raise FloatingPointError('pcg_row_partitioned: the residual is not finite (system not positive definite?)')
                if rel <= rtol:
                    break
        torch.cuda.current_stream().synchronize()
    return x, it, rel


class DeviceRows:
    """the local rows of a partitioned system on the GPU: [A_own | A_halo] stored as one square block-CSR of size
    n_loc + n_halo (the halo rows are empty), applied to torch tensors through fb_spmv_dev"""

    def __init__(self, part, indptr, data):
        import ctypes as C
        from . import _lib
        n, m = part.n_loc, part.n_loc + part.n_halo
        if n % 2 or m % 2:
            raise ValueError('DeviceRows: DoF come in (x, y) pairs')
        ip = np.concatenate((np.asarray(indptr, dtype=np.int64), np.full(m - n, int(indptr[-1]), dtype=np.int64)))
        idx = np.ascontiguousarray(part.local_cols, dtype=np.int32)
        val = np.ascontiguousarray(data, dtype=np.float64)
        self._lib, self._ctx = _lib, _lib.ctx()
        self.h = C.c_void_p()
        _lib.check(_lib.load().fb_csr_upload(self._ctx, m, _lib.ptr(ip), _lib.ptr(idx), _lib.ptr(val), 0, C.byref(self.h)))
        self.m, self.n = m, n
        self._y = None

    def stream(self):
        """the context's HIP stream as a torch stream: run the solver under ``with torch.cuda.stream(rows.stream())`` and the
        vector updates, the collectives and the SpMV are ordered on ONE stream without host synchronisation"""
        import torch
        return torch.cuda.ExternalStream(self._lib.load().fb_stream(self._ctx))

    def __call__(self, ext):
        import torch
        if self._y is None:
            self._y = torch.empty(self.m, dtype=torch.float64, device=ext.device)
        shared = torch.cuda.current_stream().cuda_stream == self._lib.load().fb_stream(self._ctx)
        if not shared:
            torch.cuda.current_stream().synchronize()             # torch's stream -> the context's stream
        self._lib.check(self._lib.load().fb_spmv_dev(self._ctx, self.h, C_ptr(ext), C_ptr(self._y)))
        if not shared:
            self._lib.check(self._lib.load().fb_sync(self._ctx))
        return self._y[:self.n]

    def free(self):
        if self.h is not None:
            self._lib.load().fb_csr_destroy(self._ctx, self.h)
            self.h = None


def C_ptr(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())
