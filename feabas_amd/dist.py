"""Sharding and the exchange steps of the multi-GPU runs.

One process per GPU.  Tile pairs / sections are independent units: every rank works on a contiguous shard -- the same
contiguous-slice partitioning the reference uses for its worker jobs (feabas/stitcher.py:375-392) -- with no collective on
the data path.  What is exchanged afterwards:
  * the variable-length match table of every rank   -> ``Exchange.gatherv`` / ``gather_match_table``
  * per-section node displacement vectors           -> ``Exchange.allgather`` / ``allgather_ragged``
and, for a coupled alignment window, the halo entries and three scalars of every PCG iteration (second half of this file).

``torch.distributed`` is the RENDEZVOUS (a gloo group: who is who, counts, the 128-byte RCCL id) and the transport of the
CPU tests.  On the GPU every byte of the data path moves through the C ABI of libfeabas_hip.so on the library's own RCCL
communicator (fb_comm_*, fb_gatherv_dev, fb_allgather_dev, fb_sendrecv_dev, fb_allreduce_f64_dev): torch never touches
the device here.
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


def _cpu():
    import torch
    return torch.device('cpu')


class Exchange:
    """The exchange steps of a sharded run: ``gatherv`` (every rank's variable-length table -> the
    root: counts first, then point-to-point transfers of exactly the bytes each rank holds -- nothing is padded and only
    the root receives) and ``allgather`` (equal contributions: node displacements).

    On the GPU box the transfers go through the C ABI (``fb_comm_*`` / ``fb_gatherv_dev`` / ``fb_allgather_dev``: one RCCL
    communicator on the context's stream, device buffers; the 128-byte communicator id travels through the
    ``torch.distributed`` group that launched the ranks).  With a gloo group (CPU tests, no GPU) the same steps run as
    ``torch.distributed`` point-to-point operations on host tensors."""

    def __init__(self, group=None, backend=None, ctx=None):
        """collective over `group`.  backend 'rccl': transfers through the C ABI on an RCCL communicator (default when a GPU
        context exists or the group's backend is nccl); 'torch': host tensors over the gloo group (CPU tests).  ctx: the
        fb_ctx whose stream carries the transfers -- default: a context of the Exchange's own, so that a gather never queues
        behind the kernels of a matcher thread and a communicator that could not be made leaves no lock behind on a context
        somebody else uses.  (The coupled-window solver needs the communicator on the context of its vectors: pass that
        context.)"""
        torch, dist = _dist()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if backend is None:
            backend = 'rccl' if dist.get_backend(group) == 'nccl' else 'torch'
        self.group = host_group(group)                       # rendezvous and host transport: never torch's GPU stack
        self.backend = backend
        self.comm = None
        self._own_ctx = None
        self.seconds = 0.0            # wall time spent inside gatherv / allgather (host view, includes the copies)
        self.bytes = 0
        if backend == 'rccl':
            import ctypes as C
            from . import _lib
            self._lib, self._C = _lib, C
            lib = _lib.load()
            if ctx is None:
                ctx = self._own_ctx = _lib.new_context()
            # every rank makes an id (the cheapest call that needs the RCCL library): a rank that cannot load it must not
            # leave the others waiting in the broadcast below, so the ranks agree on that first
            ident = np.zeros(128, dtype=np.uint8)
            rc = lib.fb_comm_unique_id(ctx, _lib.ptr(ident))
            err = None if rc == 0 else (lib.fb_last_error(ctx) or b'?').decode()
            flags = [None] * self.world
            dist.all_gather_object(flags, err, group=self.group)
            bad = [(r, e) for r, e in enumerate(flags) if e is not None]
            if bad:
                self._drop_ctx()
                raise RuntimeError(f'RCCL is not available on rank {bad[0][0]}: {bad[0][1]}')
            box = [ident.tobytes()]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            ident = np.frombuffer(box[0], dtype=np.uint8).copy()
            h = C.c_void_p()
            # ncclCommInitRank blocks until every rank has joined; a rank stuck in it is given up after a while so that the
            # caller can take another route instead of hanging the job.  The abandoned thread keeps the lock of `ctx` for as
            # long as it blocks: with a context of the Exchange's own nobody else ever waits for that lock; a caller that
            # passed its context must not use it again after this error.
            import threading
            res = {}

            def init():
                res['rc'] = lib.fb_comm_create(ctx, _lib.ptr(ident), self.rank, self.world, C.byref(h))
                res['err'] = (lib.fb_last_error(ctx) or b'').decode() if res['rc'] else ''
            th = threading.Thread(target=init, daemon=True)
            th.start()
            th.join(float(os.environ.get('FEABAS_HIP_COMM_TIMEOUT', '120')))
            if th.is_alive():
                self._own_ctx = None                         # left to the stuck thread: destroying it would block on its lock
                raise RuntimeError('fb_comm_create did not return (ncclCommInitRank waiting for the other ranks)')
            if res['rc'] != 0:
                self._drop_ctx()
                raise _lib.FeabasHipError(res['rc'], res['err'] or '?')
            self.comm, self._ctx = h, ctx

    def _drop_ctx(self):
        if self._own_ctx is not None:
            self._lib.load().fb_destroy(self._own_ctx)
            self._own_ctx = None

    def close(self):
        if self.comm is not None:
            self._lib.load().fb_comm_destroy(self._ctx, self.comm)
            self.comm = None
            self._drop_ctx()

    def rccl_ranks(self):
        """the number of ranks that really take part in the library's communicator: an all-reduce of ones through
        fb_allreduce_f64_dev (1 without a communicator)"""
        if self.comm is None:
            return 1
        _lib, lib = self._lib, self._lib.load()
        with _lib.using(self._ctx):
            d = _lib.DeviceBuffer.from_array(np.ones(1))
            _lib.check(lib.fb_allreduce_f64_dev(self._ctx, self.comm, d.ptr, d.ptr, 1, 0), h=self._ctx)
            out = int(round(float(d.to_array((1,), np.float64)[0])))
            d.free()
        return out

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
        dev = _cpu()
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
            dev = _cpu()
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
            dev = _cpu()
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
        if isinstance(ex, Exchange):
            ex.close()
    _default_exchange.clear()


def solver_comm(group=None):
    """the RCCL communicator of the coupled-window solver over `group`, on the CALLING thread's context (where the solver's
    vectors live), made on first use and kept; None when the ranks cannot have one -- a single rank, no GPU, or two ranks on
    one device (RCCL refuses that: the tests of the multi-rank logic on a one-GPU box) -- every rank reaches the same
    answer.  Collective: every rank of the group must call it."""
    torch, dist = _dist()
    from . import _lib
    hg = host_group(group)
    world = dist.get_world_size(hg)
    key = ('solver', id(group), _lib._ctx_device)
    if key in _default_exchange:
        return _default_exchange[key].comm
    force = os.environ.get('FEABAS_HIP_EXCHANGE') == 'rccl'        # a one-rank communicator: the rehearsal of the N-rank code path
    if world == 1 and not force:
        return None
    import socket
    here = (socket.gethostname(), int(os.environ.get('FEABAS_HIP_DEVICE', os.environ.get('LOCAL_RANK', '0'))), _lib._ctx is not None or _lib.gpu_available())
    seats = [None] * world
    dist.all_gather_object(seats, here, group=hg)
    if not all(s[2] for s in seats) or len({s[:2] for s in seats}) < world or os.environ.get('FEABAS_HIP_EXCHANGE') == 'host':
        return None                                          # (every rank sees the same list: the same answer everywhere)
    ex = Exchange(group, backend='rccl', ctx=_lib.ctx())
    _default_exchange[key] = ex
    return ex.comm


def allgather_ragged(arr, group=None):
    """all-gather of per-rank arrays whose first dimension differs: the list of every rank's array, on every rank.
    A gatherv to rank 0 would do for the match table (north star: one gather); this form is for callers that need the
    table everywhere (tests, small set-up tables).  ONE exchange of the counts and ONE all-gather of blocks padded to the
    largest contribution (2 collectives whatever the number of ranks; a loop of gathervs took 2 per rank)."""
    ex = exchange(group)
    arr = np.ascontiguousarray(arr)
    with ex._scope():
        cnts = ex._counts(arr.shape[0])
        cap = int(cnts.max())
        pad = np.zeros((cap,) + arr.shape[1:], dtype=arr.dtype)
        pad[:arr.shape[0]] = arr
        allp = ex._allgather(pad) if cap else np.zeros((ex.world, 0) + arr.shape[1:], dtype=arr.dtype)
    return [allp[r, :int(cnts[r])] for r in range(ex.world)]


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
#
# Who touches what: torch.distributed (a gloo group) is the RENDEZVOUS -- spans, halo lists, the RCCL id, a few host scalars --
# and the transport of the CPU tests.  On the GPU the vectors live in fb_malloc buffers, the loop is fb_cgcg_solve_dev, the halo
# travels through fb_sendrecv_dev and the scalars through fb_allreduce_f64_dev on the library's own RCCL communicator: one
# ROCm stack touches the device.
def host_group(group=None):
    """the gloo group over the ranks of `group` (itself when it already is one): object collectives of an nccl group would go
    through torch's bundled ROCm stack"""
    torch, dist = _dist()
    if dist.get_backend(group) == 'gloo':
        return group
    key = ('gloo', id(group))
    if key not in _default_exchange:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        if ranks is not None and len(ranks) < dist.get_world_size():
            # dist.new_group is a collective of the DEFAULT group: called by the members of a proper sub-group only it would
            # wait for the others for ever
            raise RuntimeError('host_group: the gloo mirror of a proper sub-group of an nccl world has to be made by every rank of the '
                               'world at start-up (torch.distributed.new_group(ranks, backend="gloo")) and passed in as the group')
        _default_exchange[key] = dist.new_group(ranks=ranks, backend='gloo')
    return _default_exchange[key]


def host_sum(values, group=None):
    """sum over the ranks of a small host array (rendezvous-sized: lambdas, norms)"""
    torch, dist = _dist()
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, np.asarray(values, dtype=np.float64), group=group)
    return np.sum(parts, axis=0)


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

    def lists(self):
        """the halo lists in the form fb_cgcg_solve_dev takes: (send_peer int32 [ns], send_off int64 [ns + 1], send_idx int32,
        recv_peer int32 [nr], recv_off int64 [nr + 1])"""
        sp = np.array(sorted(self.send), dtype=np.int32)
        so = np.concatenate(([0], np.cumsum([self.send[int(d)].size for d in sp]))).astype(np.int64)
        si = np.concatenate([self.send[int(d)] for d in sp]).astype(np.int32) if sp.size else np.zeros(0, np.int32)
        rp = np.array(sorted(self.recv), dtype=np.int32)
        ro = np.array([self.recv[int(r)].start for r in rp] + ([self.recv[int(rp[-1])].stop] if rp.size else [0]), dtype=np.int64)
        return sp, so, si, rp, ro

    def exchange(self, v_loc):
        """host path (CPU tests, processes that share one GPU): the foreign entries [n_halo] of the distributed vector whose
        local part is the numpy array v_loc, through point-to-point transfers of the torch.distributed group"""
        torch, dist = _dist()
        out = np.zeros(self.n_halo, dtype=np.float64)
        if self.world == 1 or (not self.send and not self.recv):
            return out
        glob = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
        ops, bufs = [], []
        for dst, rows in sorted(self.send.items()):
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(v_loc[rows])), glob(dst), group=self.group))
        for src, sl in sorted(self.recv.items()):
            t = torch.empty(sl.stop - sl.start, dtype=torch.float64)
            bufs.append((sl, t))
            ops.append(dist.P2POp(dist.irecv, t, glob(src), group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for sl, t in bufs:
            out[sl] = t.numpy()
        return out


def _cg_scalars(t, gamma, alpha, first):
    """the scalar recurrences of the Chronopoulos-Gear iteration (cgcg_scalars_kernel restated for the host path): a step with
    a non-positive denominator is dropped and the step after it restarts the recurrence (beta = 0) from the current iterate"""
    g_new, delta = float(t[0]), float(t[1])
    if first or alpha == 0.0:
        return g_new, (g_new / delta if delta > 0 else 0.0), 0.0, (not delta > 0) and g_new != 0.0
    beta = g_new / gamma if gamma != 0 else 0.0
    den = delta - beta * g_new / alpha
    if den > 0:
        return g_new, g_new / den, beta, False
    return g_new, 0.0, 0.0, g_new != 0.0


def pcg_row_partitioned(part, spmv, b_loc, minv_loc, rtol=1e-7, maxiter=10000, check_every=8):
    """Jacobi-preconditioned conjugate gradients on a row-partitioned system, HOST path (numpy vectors, the exchange steps
    over the torch.distributed group of `part`): the CPU statement of fb_cgcg_solve_dev, used by the gloo tests and by ranks
    that cannot have an RCCL communicator (processes sharing one GPU).

    spmv(u_ext) -> (A u)_loc for u_ext = [u_loc | u_halo] (numpy, n_loc + n_halo).  One halo exchange + one SpMV + ONE
    reduction of 3 scalars per iteration.  A step with a non-positive denominator (loss of positive definiteness in floating
    point, an exactly converged residual) is dropped and the recurrence restarts from the current iterate; `check_every`
    dropped steps in a row raise.  Stops at ||r|| <= rtol ||b||.  Returns (x_loc, iterations, relative residual)."""
    b = np.asarray(b_loc, dtype=np.float64); minv = np.asarray(minv_loc, dtype=np.float64)
    n = part.n_loc
    x = np.zeros(n); r = b.copy(); p = np.zeros(n); s = np.zeros(n)

    def product(u):
        w = np.asarray(spmv(np.concatenate((u, part.exchange(u)))), dtype=np.float64)
        return w, host_sum([r @ u, w @ u, r @ r], group=part.group) if part.world > 1 else np.array([r @ u, w @ u, r @ r])
    u = minv * r
    w, t = product(u)
    bnorm2 = float(t[2])
    if bnorm2 == 0.0 or maxiter == 0:
        return x, 0, 0.0
    gamma, alpha, beta, _ = _cg_scalars(t, 0.0, 0.0, True)
    it, rel, dropped = 0, 1.0, 0
    while it < maxiter:
        p = u + beta * p
        s = w + beta * s
        x += alpha * p
        r -= alpha * s
        u = minv * r
        w, t = product(u)
        it += 1
        rel = (float(t[2]) / bnorm2) ** 0.5
        if not np.isfinite(rel):
            raise FloatingPointError('pcg_row_partitioned: the residual is not finite (system not positive definite?)')
        if rel <= rtol:
            break
        gamma, alpha, beta, drop = _cg_scalars(t, gamma, alpha, False)
        dropped = dropped + 1 if drop else 0
        if dropped >= check_every:
            raise FloatingPointError(f'pcg_row_partitioned: {dropped} steps in a row were dropped (p^T A p <= 0): the system is not positive definite')
    return x, it, rel


class DeviceRows:
    """the local rows of a partitioned system on the GPU: [A_own | A_halo] stored as one square block-CSR of size
    n_loc + n_halo (the halo rows are empty), applied to device vectors through fb_spmv_dev / fb_cgcg_solve_dev"""

    def __init__(self, part, indptr, data):
        import ctypes as C
        from . import _lib
        n, m = part.n_loc, part.n_loc + part.n_halo
        self._lib, self._ctx = _lib, _lib.ctx()
        self.h, self.m, self.n = None, m, n
        if n % 2 or m % 2:
            raise ValueError('DeviceRows: DoF come in (x, y) pairs')
        if m == 0:
            return                                          # a rank that owns nothing (more ranks than sections)
        ip = np.concatenate((np.asarray(indptr, dtype=np.int64), np.full(m - n, int(indptr[-1]), dtype=np.int64)))
        idx = np.ascontiguousarray(part.local_cols, dtype=np.int32)
        val = np.ascontiguousarray(data, dtype=np.float64)
        self.h = C.c_void_p()
        _lib.check(_lib.load().fb_csr_upload(self._ctx, m, _lib.ptr(ip), _lib.ptr(idx), _lib.ptr(val), 0, C.byref(self.h)))

    def free(self):
        if self.h is not None:
            self._lib.load().fb_csr_destroy(self._ctx, self.h)
            self.h = None


def pcg_row_partitioned_dev(part, rows, b_loc, minv_loc, rtol=1e-7, maxiter=10000, check_every=8, comm=None):
    """the coupled-window PCG on the GPU.  b_loc / minv_loc: numpy [n_loc]; rows: DeviceRows; comm: the fb_comm of the ranks of
    `part` (an ``Exchange(...).comm`` made on the CALLING thread's context) or None.

    With a communicator, or on a single rank, the whole loop is ONE C call (fb_cgcg_solve_dev): vectors in fb_malloc buffers,
    halo through fb_sendrecv_dev, scalars through fb_allreduce_f64_dev, the host reads 64 bytes every `check_every` iterations.
    Several ranks without a communicator (processes that share one GPU -- RCCL refuses two ranks on a device -- as in the
    tests of this module) run the same kernels step by step and carry halo and scalars over the host group.
    Returns (x_loc numpy, iterations, relative residual)."""
    import ctypes as C
    from . import _lib
    lib, ctx = _lib.load(), rows._ctx
    n, nh = part.n_loc, part.n_halo
    b = np.ascontiguousarray(b_loc, dtype=np.float64); minv = np.ascontiguousarray(minv_loc, dtype=np.float64)
    limit = 100000 if maxiter is None or maxiter < 0 else int(maxiter)
    d_b = _lib.DeviceBuffer.from_array(b) if n else _lib.DeviceBuffer(16)
    d_minv = _lib.DeviceBuffer.from_array(minv) if n else _lib.DeviceBuffer(16)
    d_x = _lib.DeviceBuffer(8 * max(n, 2))
    bufs = [d_b, d_minv, d_x]
    try:
        if comm is not None or part.world == 1:
            sp, so, si, rp, ro = part.lists()
            d_si = _lib.DeviceBuffer.from_array(si) if si.size else None
            if d_si is not None:
                bufs.append(d_si)
            it, rel, bn = C.c_int(), C.c_double(), C.c_double()
            rc = lib.fb_cgcg_solve_dev(ctx, comm, rows.h, n, nh, d_b.ptr, d_minv.ptr, d_x.ptr, sp.size, _lib.ptr(sp), _lib.ptr(so),
                                       d_si.ptr if d_si is not None else None, rp.size, _lib.ptr(rp), _lib.ptr(ro), float(rtol), limit,
                                       int(check_every), C.byref(it), C.byref(rel), C.byref(bn))
            if rc == _lib.FB_ERR_BREAKDOWN:
                raise FloatingPointError((lib.fb_last_error(ctx) or b'?').decode())
            _lib.check(rc, allow=(_lib.FB_ERR_NOCONV,), h=ctx)
            return (d_x.to_array((n,), np.float64) if n else np.zeros(0)), it.value, rel.value
        # ---- ranks without a communicator: the same kernels, halo and scalars over the host group
        m = n + nh
        d_ext, d_y = _lib.DeviceBuffer(8 * max(m, 2)), _lib.DeviceBuffer(8 * max(m, 2))
        d_vec = [_lib.DeviceBuffer(8 * max(n, 1)) for _ in range(4)]          # r, p, s, w
        d_state, d_t3, d_scr = _lib.DeviceBuffer(64), _lib.DeviceBuffer(32), _lib.DeviceBuffer(8 * 3 * 1024)
        bufs += [d_ext, d_y, d_state, d_t3, d_scr] + d_vec
        d_r, d_p, d_s, d_w = d_vec
        for d in (d_ext, d_y, d_state, d_x, d_p, d_s, d_w):
            _lib.check(lib.fb_memset(ctx, d.ptr, 0, d.nbytes), h=ctx)
        if n:
            _lib.check(lib.fb_memcpy_d2d(ctx, d_r.ptr, d_b.ptr, 8 * n), h=ctx)

        def product_and_dots(first):
            u = d_ext.to_array((n,), np.float64) if n else np.zeros(0)
            halo = part.exchange(u)
            if nh:
                _lib.check(lib.fb_memcpy_h2d(ctx, d_ext.offset(8 * n), _lib.ptr(halo), 8 * nh), h=ctx)
            if m:
                _lib.check(lib.fb_spmv_dev(ctx, rows.h, d_ext.ptr, d_y.ptr), h=ctx)
            if n:
                _lib.check(lib.fb_memcpy_d2d(ctx, d_w.ptr, d_y.ptr, 8 * n), h=ctx)
            _lib.check(lib.fb_cgcg_dots_dev(ctx, n, d_r.ptr, d_ext.ptr, d_w.ptr, d_scr.ptr, d_t3.ptr), h=ctx)
            t3 = host_sum(d_t3.to_array((3,), np.float64), group=part.group)
            _lib.check(lib.fb_memcpy_h2d(ctx, d_t3.ptr, _lib.ptr(np.ascontiguousarray(t3)), 24), h=ctx)
            _lib.check(lib.fb_cgcg_scalars_dev(ctx, d_t3.ptr, d_state.ptr, 1 if first else 0), h=ctx)
            return d_state.to_array((8,), np.float64)
        upd = lambda: _lib.check(lib.fb_cgcg_update_dev(ctx, n, d_state.ptr, d_minv.ptr, d_x.ptr, d_r.ptr, d_ext.ptr, d_w.ptr, d_p.ptr, d_s.ptr), h=ctx)
        upd()                                                   # alpha = beta = 0, p = s = 0: u = minv r
        st = product_and_dots(True)
        bnorm2 = float(st[3])
        if bnorm2 == 0.0 or limit == 0:
            return (d_x.to_array((n,), np.float64) if n else np.zeros(0)), 0, 0.0
        it, rel, seen = 0, 1.0, 0.0
        while it < limit:
            upd()
            st = product_and_dots(False)
            it += 1
            rel = (float(st[3]) / bnorm2) ** 0.5
            if not np.isfinite(rel):
                raise FloatingPointError('pcg_row_partitioned_dev: the residual is not finite (system not positive definite?)')
            if rel <= rtol:
                break
            if it % check_every == 0:
                if st[4] - seen >= check_every:
                    raise FloatingPointError(f'pcg_row_partitioned_dev: every step of the last {check_every} was dropped (p^T A p <= 0)')
                seen = st[4]
        return (d_x.to_array((n,), np.float64) if n else np.zeros(0)), it, rel
    finally:
        for d in bufs:
            d.free()
