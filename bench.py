#!/usr/bin/env python3
"""Benchmark of the FEABAS hot paths on MI355X (contract: see the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Headline metric (BASELINE.json): tile-pair NCC matches/s on config[1] -- 1024 synthetic
4096x4096 tile pairs (512 left-right + 512 up-down, 10 % overlap + margin 100 => strips of
4096x510 / 510x4096 uint8), inputs resident in HBM.  One "step" = one pass of the device pair
matcher (feabas_amd/stitch_pipeline.py) over one batch of --pairs-per-step pairs.  The same
JSON line carries the FEM half of the metric ("fem": PCG iterations/s on the 1e6-DoF system of
config[2]), the roofline of the dominant kernel and a CPU baseline timed with the oracle.
With N > 1 every rank matches its own shard of pairs (no data-path collective) and the match
tables of a step meet on rank 0 through ONE gather (counts, then point-to-point transfers over
RCCL: fb_gatherv_dev); value = all pairs / max-over-ranks time.  `--gpus N` without a launcher
environment starts the N ranks itself (python -m torch.distributed.run as a child process, before
anything in this process touches the GPU).  Beside the headline the line carries the two 8-GPU
workloads of BASELINE.json scaled to the number of ranks (weak scaling): `stitch_sections`
(config[3]: 8 sections x 400 tiles per rank = 760 edge + 722 corner pairs per section, one gather
of the match table) and `align_sections` (config[4]: 16 sections x 250 k nodes per rank, every
section relaxed against its locked neighbours, one all-gather of the node displacements).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
RESIDUE_LEN = 2.0              # configs/default_stitching_configs.yaml:15 (matcher_config.residue_len of the default stitching run)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=48)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--pairs-per-step', type=int, default=512, help='tile pairs per step = one batch of the hot path; processed as sub-batches of --sub-batch pairs dealt to the host threads')
    ap.add_argument('--sub-batch', type=int, default=128, help='pairs per StripBatchMatcher call (the per-call host bookkeeping is shared by more pairs; 64 / 128 / 256 give the same rate on a fast host)')
    ap.add_argument('--resident-pairs', type=int, default=1024)
    ap.add_argument('--tile', type=int, default=4096)
    ap.add_argument('--host-threads', type=int, default=8, help='host threads driving the device (even; steps are dealt round-robin, LR and UD batches alternate)')
    ap.add_argument('--multi-stream', type=int, default=1, help='1: one context (HIP stream) per host thread; 0: all threads share one stream')
    ap.add_argument('--warp', type=float, default=0.4, help='amplitude (px) of the smooth sub-pixel warp between the strips of a pair (SURVEY config 2)')
    ap.add_argument('--host-ingest-threads', type=int, default=8)
    ap.add_argument('--host-ingest-batch', type=int, default=32, help='pairs per chunk of the PCIe-inclusive measurement (a chunk is packed, copied and matched by one host thread)')
    ap.add_argument('--host-ingest-pairs', type=int, default=1024, help='pairs of the PCIe-inclusive measurement (0: skip)')
    ap.add_argument('--no-fem', action='store_true')
    ap.add_argument('--no-align', action='store_true', help='skip the alignment-side block matcher section')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-xcorr-classes', action='store_true', help='skip the kernel-only xcorr_fft rates (the PMC passes of tools/profile_round.sh: only the shapes of the timed steps are launched)')
    ap.add_argument('--no-deformed', action='store_true', help='skip the 2-px-warp sub-record (deformed-mesh branch)')
    ap.add_argument('--fem-grid', type=int, default=708)
    ap.add_argument('--fem-iters', type=int, default=200)
    ap.add_argument('--fem-cpu-iters', type=int, default=40, help='iterations of the CPU baseline PCG (about 0.1 s each at 1e6 DoF)')
    ap.add_argument('--stitch-sections', type=int, default=8, help='config[3]: sections of 400 tiles PER RANK (64 sections on 8 GPUs); 0: skip')
    ap.add_argument('--align-sections', type=int, default=16, help='config[4]: sections of --align-grid^2 nodes PER RANK (128 sections on 8 GPUs); 0: skip')
    ap.add_argument('--align-threads', type=int, default=4, help='host threads (one context and one SLM each) that share the sections of config[4]')
    ap.add_argument('--align-grid', type=int, default=500, help='nodes per side of a section mesh of config[4] (500 x 500 = 250 k nodes)')
    ap.add_argument('--cpu-pool-seconds', type=float, default=20.0, help='wall-clock budget of the all-core CPU baseline (process pool); 0: skip')
    ap.add_argument('--allow-host-exchange', action='store_true', help='N > 1: if the RCCL communicator of the C ABI cannot be made, run the exchange steps over the gloo group on host arrays instead of exiting non-zero')
    ap.add_argument('--dry-run', action='store_true', help='no GPU work: the ranks exercise sharding and the exchange steps on synthetic tables over gloo (CPU tests)')
    return ap.parse_args()


def spawn_ranks(args):
    """`--gpus N` (N > 1) outside a launcher: start the N ranks as children of `python -m torch.distributed.run` and pass
    rank 0's JSON line through.  Nothing here initialises the GPU (torch.cuda.device_count() does not on this image); a
    child failure is this process's exit code."""
    import socket
    import subprocess
    if not args.dry_run:
        try:
            import torch
            ndev = torch.cuda.device_count()
        except Exception:                                 # noqa: BLE001
            ndev = 0
        if ndev < args.gpus:
            sys.stderr.write(f'bench.py: --gpus {args.gpus} asks for {args.gpus} ranks (one process per GPU) but {ndev} GPU(s) are visible on this host\n')
            sys.exit(2)
    sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    sys.exit(subprocess.call(cmd, env=env))


def ncc_bytes_model(H, W, spacings_blocks):
    """Algorithmic HBM bytes of one tile pair: strips read once, DoG outputs
    written once, every NCC = inputs + 48 S (streaming class) or inputs + 20 (on-chip class)."""
    from feabas_amd.matcher import next_fast_len as nfl
    total = 2 * H * W                                  # uint8 strips
    total += 2 * (H // 2) * (W // 2) * 4 + 2 * H * W * 4   # coarse + fine DoG writes
    hc, wc = H // 2, W // 2
    fh, fw = nfl(2 * hc - 1), nfl(2 * wc - 1)
    total += 2 * hc * wc * 4 + 48 * fh * (fw // 2 + 1)
    for (nblk, h, w, pad) in spacings_blocks:
        fh, fw = (nfl(2 * h - 1), nfl(2 * w - 1)) if pad else (nfl(h), nfl(w))
        if fh * fw <= 256 * 256:
            total += nblk * (2 * h * w * 4 + 20)
        else:
            total += nblk * (2 * h * w * 4 + 48 * fh * (fw // 2 + 1))
    return total


def build_fem_system(grid, nlinks, seed=0):
    """config[2]: grid x grid node mesh (h = 10), one locked twin, random links, smooth imposed displacement."""
    from feabas_amd import mesh, optimizer
    n = grid
    xs = 10.0 * np.arange(n)
    vx, vy = np.meshgrid(xs, xs)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(n * n).reshape(n, n)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel(); c = idx[1:, :-1].ravel(); d = idx[1:, 1:].ravel()
    par = ((np.arange(n - 1)[None, :] + np.arange(n - 1)[:, None]) % 2).ravel().astype(bool)
    t0 = np.where(par[:, None], np.stack((a, b, c), -1), np.stack((a, b, d), -1))
    t1 = np.where(par[:, None], np.stack((b, d, c), -1), np.stack((a, d, c), -1))
    tri = np.concatenate((t0, t1), axis=0).astype(np.int32)
    rng = np.random.default_rng(seed)
    L = 10.0 * (n - 1)
    disp = np.stack((5 * np.sin(2 * np.pi * v[:, 1] / L), 4 * np.cos(2 * np.pi * v[:, 0] / L)), axis=-1)
    m0 = mesh.Mesh(v + disp, tri, uid=0, locked=True)
    m1 = mesh.Mesh(v.copy(), tri, uid=1)
    tid = rng.integers(0, tri.shape[0], nlinks)
    B = rng.dirichlet((1, 1, 1), nlinks)
    w = rng.uniform(0.3, 1.0, nlinks).astype(np.float32)
    link = optimizer.Link(m0, m1, tid, tid, B, B, weight=w)
    return optimizer.SLM([m0, m1], [link], stiffness_lambda=1.0, crosslink_lambda=-1.0)


def bench_fem(args, lib, ctx, _lib):
    """PCG iterations/s and time to 1e-4 on the ~1e6-DoF relaxation (config[2])."""
    t0 = time.time()
    slm = build_fem_system(args.fem_grid, 200000)          # the inputs: numpy grid, random matches, Mesh / Link / SLM objects
    t_inputs = time.time() - t0
    t0 = time.time()
    slm._assemble(0, 1, 1)                                 # symbolic (pattern on the device) + numeric assembly
    _lib.check(lib.fb_sync(ctx))
    t_asm_first = time.time() - t0
    t0 = time.time()
    slm._assemble(0, 1, 1)                                 # numeric only (pattern cached)
    _lib.check(lib.fb_sync(ctx))
    t_asm = time.time() - t0
    sl, cl = slm.relative_lambda_trace(1.0, -1.0)
    _lib.check(lib.fb_sys_form(ctx, slm._sys, sl, cl))
    nv = C.c_int64(); nnzb = C.c_int64(); nl = C.c_int64()
    _lib.check(lib.fb_sys_info(ctx, slm._sys, C.byref(nv), C.byref(nnzb), C.byref(nl)))
    rr = C.c_double()
    _lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, 20, C.byref(rr)))          # warm-up
    _lib.check(lib.fb_sync(ctx))
    t0 = time.time()
    _lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, args.fem_iters, C.byref(rr)))
    _lib.check(lib.fb_sync(ctx))
    dt = time.time() - t0
    # per-kernel durations from a second, event-bracketed run (the events cost a few us per launch)
    _lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
    _lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, 64, C.byref(rr)))
    _lib.check(lib.fb_sync(ctx))
    _lib.check(lib.fb_prof_enable(ctx, 0))
    prof = _lib.prof_snapshot()
    # time to 1e-4 relative residual (the reference's final_elastic tolerance, stitching_configs.yaml:63-72)
    # (first call apart: round 2 timed a single call, which paid ~35 ms for the first host buffer of that size the runtime had
    # to map -- every host <-> device copy now goes through the context's pinned staging ring, fb_copy_h2d / fb_copy_d2h)
    x = np.zeros(2 * nv.value)
    it = C.c_int()
    t_calls = []
    for _ in range(4):
        x[:] = 0
        t0 = time.time()
        _lib.check(lib.fb_sys_solve(ctx, slm._sys, _lib.ptr(x), 0, 1e-4, 0.0, -1, 1, C.byref(it), C.byref(rr)))
        t_calls.append(time.time() - t0)
    t_solve_first, t_solve = t_calls[0], float(np.median(t_calls[1:]))
    n = 2 * nv.value
    # bytes per PCG iteration with this storage: 2x2 blocks of float64 + int32 block column + int32 row pointer,
    # SpMV vector traffic (x gather counted once, y write) + 12 vector passes (SURVEY.md sec.8d formula, BSR sizes)
    it_bytes = nnzb.value * (32 + 4) + nv.value * 4 + n * 8 * 2 + 12 * n * 8
    k1 = prof.get('pcg_spmv_fused', (0, 0.0, 0.0)); k2 = prof.get('pcg_update_fused', (0, 0.0, 0.0))
    spmv_bytes = nnzb.value * 36 + nv.value * 4 + n * 8 * 5          # A + z,p_old own rows + p_new, Ap writes (+gather once)
    out = dict(dof=n, nnz_blocks=nnzb.value, links=nl.value, iters_per_s=args.fem_iters / dt,
               ms_per_iter=1e3 * dt / args.fem_iters, bytes_per_iter=it_bytes,
               hbm_gbs=it_bytes * args.fem_iters / dt / 1e9, hbm_frac=it_bytes * args.fem_iters / dt / 1e9 / HBM_PEAK_GBS,
               solve_to_1e4_s=t_solve, solve_to_1e4_first_call_s=t_solve_first, solve_iters=it.value, solve_relres=rr.value,
               solve_to_1e4_device_s=(it.value + 32 - it.value % 32) * 1e-3 * (k1[1] / max(k1[0], 1) + k2[1] / max(k2[0], 1)),
               assemble_numeric_s=t_asm, assemble_first_s=t_asm_first, build_inputs_s=t_inputs,
               spmv_kernel_us=1e3 * k1[1] / max(k1[0], 1), update_kernel_us=1e3 * k2[1] / max(k2[0], 1),
               spmv_kernel_gbs=spmv_bytes / max(1e-9, (k1[1] / max(k1[0], 1)) * 1e-3) / 1e9, x=x)
    if not args.no_cpu_baseline and getattr(args, '_solo', True):
        # reported baseline (rank 0 at N = 1 only): the same A, b through the oracle's Jacobi-PCG on scipy CSR, one host thread
        # (scipy's SpMV is single threaded), a bounded number of iterations
        from feabas_amd.mesh import bsr_download
        from oracle import fem_ref
        A = bsr_download(slm._sys, 4, nv.value, nnzb.value)
        b = np.empty(n)
        _lib.check(lib.fb_sys_get(ctx, slm._sys, 5, _lib.ptr(b)))
        t0 = time.time()
        _, cit, _ = fem_ref.pcg(A, b, rtol=1e-30, maxiter=args.fem_cpu_iters)
        cdt = time.time() - t0
        out['cpu_baseline'] = dict(value=cit / cdt, unit='PCG iterations/s', cores=1, kind='port',
                                   sample=f'{cit} iterations of oracle/fem_ref.pcg (scipy CSR, {A.nnz} non-zeros) on the same system, {cdt:.1f} s; '
                                          'scipy SpMV is single threaded, so this is also the all-core figure of the reference path')
        out['cpu_baseline']['scipy'] = fem_cpu_krylov(A, b)
        # the 2 x 2 blocks store explicit zeros (on the nu = 0 structured grid of this system almost half of the scalars): the same
        # figures with the bytes of those zeros taken out -- what a scalar-exact storage would have to move per iteration
        zeros = 4 * nnzb.value - int(A.nnz)
        out['stored_scalars'] = 4 * nnzb.value
        out['nonzero_scalars'] = int(A.nnz)
        out['useful_bytes_per_iter'] = it_bytes - 8 * zeros
        out['hbm_frac_zeros_excluded'] = (it_bytes - 8 * zeros) * args.fem_iters / dt / 1e9 / HBM_PEAK_GBS
    # hard variant: 5 k links, about one per 100 nodes -- the elastic term carries the solution
    # across the mesh; solved to 1e-4 like the main system and to the reference's default tolerance 1e-7
    # (default_alignment_configs.yaml: elastic_params.tol), where the smooth modes cost thousands of iterations
    del slm
    hard = build_fem_system(args.fem_grid, 5000, seed=1)
    hard._assemble(0, 1, 1)
    sl, cl = hard.relative_lambda_trace(1.0, -1.0)
    _lib.check(lib.fb_sys_form(ctx, hard._sys, sl, cl))
    xh = np.zeros(n)
    t0 = time.time()
    rc = lib.fb_sys_solve(ctx, hard._sys, _lib.ptr(xh), 0, 1e-4, 0.0, 20000, 1, C.byref(it), C.byref(rr))
    out['hard_5k_links'] = dict(solve_to_1e4_s=time.time() - t0, solve_iters=it.value, solve_relres=rr.value, converged=bool(rc == 0))
    xh[:] = 0
    t0 = time.time()
    rc = lib.fb_sys_solve(ctx, hard._sys, _lib.ptr(xh), 0, 1e-7, 0.0, 200000, 1, C.byref(it), C.byref(rr))
    out['hard_5k_links'].update(solve_to_1e7_s=time.time() - t0, solve_to_1e7_iters=it.value, solve_to_1e7_relres=rr.value, solve_to_1e7_converged=bool(rc == 0))
    del hard
    # few links: the mesh is pinned at 50 points only and the smooth modes between them are what the Jacobi-PCG has to work for
    sparse_ = build_fem_system(args.fem_grid, 50, seed=2)
    sparse_._assemble(0, 1, 1)
    sl, cl = sparse_.relative_lambda_trace(1.0, -1.0)
    _lib.check(lib.fb_sys_form(ctx, sparse_._sys, sl, cl))
    xh[:] = 0
    t0 = time.time()
    rc = lib.fb_sys_solve(ctx, sparse_._sys, _lib.ptr(xh), 0, 1e-7, 0.0, 400000, 1, C.byref(it), C.byref(rr))
    dt_s = time.time() - t0
    out['hard_50_links'] = dict(solve_to_1e7_s=dt_s, solve_to_1e7_iters=it.value, solve_to_1e7_relres=rr.value, converged=bool(rc == 0),
                                iters_per_s=it.value / max(dt_s, 1e-9))
    # the same system with the aggregation multigrid as the preconditioner (precond 2: what the reference asks pyamg's
    # smoothed_aggregation for, optimizer.py:1962-1971); the time includes building the hierarchy (aggregates and coarse
    # patterns on the host, Galerkin products on the device) -- second call: the allocation cache is warm
    mgr = []
    for _ in range(2):
        xh[:] = 0
        t0 = time.time()
        rc = lib.fb_sys_solve(ctx, sparse_._sys, _lib.ptr(xh), 0, 1e-7, 0.0, 400000, 2, C.byref(it), C.byref(rr))
        mgr.append((time.time() - t0, it.value, rr.value, rc))
    out['hard_50_links']['multigrid'] = dict(solve_to_1e7_s=mgr[-1][0], first_call_s=mgr[0][0], solve_to_1e7_iters=mgr[-1][1], solve_to_1e7_relres=mgr[-1][2],
                                              converged=bool(mgr[-1][3] == 0), note='set-up of the hierarchy inside the time; V(1,1) cycle per iteration')
    # precond 3 ('auto'): Jacobi-PCG for the iterations a multigrid solve costs, then the multigrid-PCG from the iterate reached
    xh[:] = 0
    t0 = time.time()
    rc = lib.fb_sys_solve(ctx, sparse_._sys, _lib.ptr(xh), 0, 1e-7, 0.0, 400000, 3, C.byref(it), C.byref(rr))
    out['hard_50_links']['auto'] = dict(solve_to_1e7_s=time.time() - t0, solve_to_1e7_iters=it.value, solve_to_1e7_relres=rr.value, converged=bool(rc == 0))
    del sparse_
    return out


def bench_xcorr_classes(lib, ctx, _lib, only=None):
    """kernel-only rate of matcher.xcorr_fft at the three shape classes of the 4k configuration:
    block pairs per second with the stacks resident in HBM"""
    out = {}
    rng = np.random.default_rng(1)
    for name, (N, h, w, pad, sub) in {'fine_75x73_fft75x75': (24640, 75, 73, 0, 1), 'coarse_1024x510_fft2048x1024': (128, 1024, 510, 1, 0),
                                      'global_2048x255_fft4096x512': (32, 2048, 255, 1, 0),
                                      # the global strip when the coarse level runs at full resolution (coarse_downsample 1): columns of 8192 points
                                      'global_fullres_4096x510_fft8192x1024': (32, 4096, 510, 1, 0),
                                      # alignment block classes (default_alignment_configs.yaml:16-23: spacings [400, 100] x 0.7) and the
                                      # README stitching grid, all padded: compile-time mixed-radix streaming kernels (fb_ncc_ct.hip)
                                      'align_280x280_fft576x576': (1024, 280, 280, 1, 1), 'align_70x70_fft144x144': (1024, 70, 70, 1, 1),
                                      'readme_74x72_fft150x144_run_at_160x144': (1024, 74, 72, 1, 1), 'readme_67x75_fft135x150_run_at_144x160': (1024, 67, 75, 1, 1),
                                      # literal stress variant of SURVEY.md sec.8d: whole 4096 x 4096 tiles
                                      'full_4096x4096_fft4096x4096': (4, 4096, 4096, 0, 1), 'full_4096x4096_fft8192x8192_padded': (4, 4096, 4096, 1, 1)}.items():
        if only and not any(o_ in name for o_ in only):
            continue
        a = rng.standard_normal((min(N, 256), h, w)).astype(np.float32)
        a = np.tile(a, (-(-N // a.shape[0]), 1, 1))[:N]
        d0 = _lib.DeviceBuffer.from_array(a); d1 = _lib.DeviceBuffer.from_array(np.roll(a, (2, -3), (1, 2)))
        o = _lib.DeviceBuffer(N * 20)
        ms = C.c_float(); best = 1e9
        for r in range(4):
            _lib.check(lib.fb_timer_start(ctx))
            _lib.check(lib.fb_ncc_batch_dev(ctx, d0.ptr, d1.ptr, N, 1, h, w, h, w, pad, sub, 2, o.ptr, o.offset(8 * N), o.offset(16 * N)))
            _lib.check(lib.fb_timer_stop(ctx, C.byref(ms)))
            if r:
                best = min(best, ms.value)
        out[name] = dict(block_pairs_per_s=N / (best * 1e-3), us_per_block_pair=1e3 * best / N)
        if name.startswith('full_'):
            # whole tiles: the streaming passes at their algorithmic bytes (rows read both images and write T, the
            # column pass reads T and writes V, the inverse-row pass reads V; MIRROR confidence: V has two planes), against 8 TB/s
            F = 8192 if pad else 4096
            sw = F // 2 + 1
            nbytes = 8.0 * h * w + sw * (2 * 16.0 * h + 2 * 8.0 * 2 * F)
            out[name].update(fft_shape=[F, F], algorithmic_bytes_per_pair=nbytes, algorithmic_gbs=nbytes * N / (best * 1e-3) / 1e9,
                             frac_of_8tbs=nbytes * N / (best * 1e-3) / 8e12)
        for b in (d0, d1, o):
            b.free()
    return out


def bench_align(lib, ctx, _lib, S=8192, mesh_size=50.0, nblocks=512, B=280):
    """alignment side: matcher.bboxes_mesh_renderer_matcher through two general triangulated
    meshes (smooth non-affine fields) over S x S uint8 sections resident in HBM: render both block stacks, masked DoG, NCC"""
    from feabas_amd import constant as const
    from feabas_amd import matcher as fmatcher
    from feabas_amd import renderer
    from feabas_amd.mesh import Mesh
    rng = np.random.default_rng(3)
    img = rng.integers(0, 255, (S, S), dtype=np.uint8)
    images = [renderer.ResidentImage(img), renderer.ResidentImage(np.roll(img, (3, -5), axis=(0, 1)))]
    n = int((S - 1) / mesh_size) + 1
    gx, gy = np.meshgrid(np.linspace(0, S - 1, n), np.linspace(0, S - 1, n))
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1))).astype(np.int32)
    rends = []
    for k in range(2):
        v = np.stack((gx.ravel(), gy.ravel()), axis=-1)
        inner = ((gx > 0) & (gx < S - 1) & (gy > 0) & (gy < S - 1)).ravel()
        v[inner] += rng.uniform(-0.25, 0.25, (int(inner.sum()), 2)) * mesh_size
        M = Mesh(v, tris)
        sxy = v / S
        M.set_vertices(v + 4.0 * np.stack((np.sin(3.1 * sxy[:, 1] + k), np.cos(2.3 * sxy[:, 0] - k)), axis=-1), const.MESH_GEAR_MOVING)
        rends.append(renderer.MeshRenderer.from_mesh(M, image_loader=images[k]))
    x0 = rng.integers(0, S - B, nblocks); y0 = rng.integers(0, S - B, nblocks)
    bboxes = np.stack((x0, y0, x0 + B, y0 + B), axis=-1)
    best = np.inf
    for _ in range(3):
        t0 = time.time()
        xy0, xy1, conf = fmatcher.bboxes_mesh_renderer_matcher(rends[0].mesh, rends[1].mesh, rends[0], rends[1], bboxes, bboxes, sigma=2.5)
        best = min(best, time.time() - t0)
    for r in rends:
        r.free()
    for im in images:
        im.free()
    return dict(value=nblocks / best, unit='blocks/s', blocks=nblocks, block=[B, B], triangles=int(tris.shape[0]), image=[S, S], sigma=2.5,
                median_conf=float(np.median(conf)), ms_per_call=1e3 * best,
                note='best of 3 calls; both stacks rendered through their meshes from resident uint8 sections (exact piecewise-linear tier), '
                     'masked DoG, padded NCC; results (dx, dy, conf per block) returned to the host')


def bench_section_matcher(lib, ctx, _lib, S=8192, mesh_size=100.0, reps=3, config=None):
    """alignment side end to end (align_main --mode matching): ONE pair of S x S uint8 sections through
    matcher.section_matcher -- two irregular meshes (both free: the floating system of matcher.py:551), spacings 280 / 70 px
    (0.7 x the [400, 100] of alignment_configs.yaml:16-23), render + DoG + NCC + relaxation per round -- against the smooth
    field the second section was resampled through"""
    from scipy.ndimage import gaussian_filter, map_coordinates
    from scipy.spatial import Delaunay
    from feabas_amd import matcher as fmatcher
    from feabas_amd import renderer
    from feabas_amd.mesh import Mesh
    rng = np.random.default_rng(0)
    t = gaussian_filter(rng.standard_normal((S, S)).astype(np.float32), 1.5)
    t += 0.7 * t.std() * gaussian_filter(rng.standard_normal((S, S)).astype(np.float32), 12) / 0.03
    base = np.clip(128 + 40 * t / t.std(), 0, 255).astype(np.uint8)
    del t
    yy, xx = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32), indexing='ij')

    def field(x, y):
        return (8.0 * np.sin(2 * np.pi * y / (0.8 * S) + 0.4) + 3.0 * (x / S) ** 2, 6.0 * np.cos(2 * np.pi * x / (0.7 * S)) - 2.0 * (x / S) * (y / S))
    ux, uy = field(xx, yy)
    img1 = np.clip(np.rint(map_coordinates(base, [yy + uy, xx + ux], order=1, mode='nearest', output=np.float32)), 0, 255).astype(np.uint8)
    del yy, xx, ux, uy
    meshes = []
    for k in range(2):
        g = np.arange(0, S, mesh_size)
        gx, gy = np.meshgrid(np.append(g, S - 1), np.append(g, S - 1))
        v = np.stack((gx.ravel(), gy.ravel()), axis=-1).astype(np.float64)
        inner = (v[:, 0] > 0) & (v[:, 0] < S - 1) & (v[:, 1] > 0) & (v[:, 1] < S - 1)
        v[inner] += rng.uniform(-0.3, 0.3, (int(inner.sum()), 2)) * mesh_size
        meshes.append(Mesh(v, Delaunay(v).simplices.astype(np.int32), uid=k))
    images = [renderer.ResidentImage(base), renderer.ResidentImage(img1)]
    best, out = np.inf, None
    for _ in range(reps):
        m0, m1 = meshes[0].copy(), meshes[1].copy()
        trace = []
        t0 = time.time()
        kw = dict(spacings=[280, 70], conf_thresh=0.3, residue_len=3.0) if config is None else dict(config)
        xy0, xy1, w, strain = fmatcher.section_matcher(m0, m1, images[0], images[1], trace=trace, **kw)
        dt = time.time() - t0
        if dt < best:
            best, out = dt, (xy0, xy1, trace)
    # throughput: the aligner matches many section pairs (align_main --mode matching deals them to workers): T host threads with a
    # context (stream) each run the same pair concurrently -- the kernels and copies of one thread's call run while another
    # thread is in python between two entries (27 of the 56-62 ms of a call are, tools/bench_section_matcher.py --entries)
    conc = None
    if config is None:
        try:
            T, R = 4, 3
            kwj = dict(spacings=[280, 70], conf_thresh=0.3, residue_len=3.0)
            for rep in range(2):                            # first pass: per-context code objects and arenas
                jobs = [(meshes[0].copy(), meshes[1].copy(), images[0], images[1], kwj) for _ in range(T * R)]
                t0 = time.time()
                outs = fmatcher.section_matcher_batch(jobs, threads=T)
                dtc = time.time() - t0
            counts = [int(o[0] is not None) for o in outs]
            errs = []
            conc = dict(error=f'{type(errs[0]).__name__}: {errs[0]}') if errs else dict(
                value=sum(counts) / dtc, unit='section pairs/s', host_threads=T, pairs=sum(counts), seconds=dtc,
                note='matcher.section_matcher_batch: the same pair as 12 jobs on 4 host threads, one context (stream) each: throughput of the matching stage of a stack, not the latency of a pair')
        except Exception as e:                              # noqa: BLE001 -- a side record
            conc = dict(error=f'{type(e).__name__}: {e}')
    for im in images:
        im.free()
    xy0, xy1, trace = out
    ex, ey = field(xy1[:, 0], xy1[:, 1])
    err = np.hypot(xy1[:, 0] - xy0[:, 0] + ex, xy1[:, 1] - xy0[:, 1] + ey)
    return dict(value=1.0 / best, unit='section pairs/s', ms_per_pair=1e3 * best, image=[S, S], triangles=[int(m.num_triangles) for m in meshes],
                vertices=[int(m.num_vertices) for m in meshes], matches=int(xy0.shape[0]),
                median_error_px=float(np.median(err)), p95_error_px=float(np.quantile(err, 0.95)), concurrent=conc,
                rounds=[dict(blocks=int(r['blocks']), kept=int(r['kept']), max_dis=float(r['max_dis']), solve_iters=r.get('solve', {}).get('iters'),
                             precond=r.get('solve', {}).get('precond')) for r in trace],
                config='spacings [280, 70], conf_thresh 0.3, residue_len 3, the other keywords at section_matcher\'s defaults' if config is None else config,
                note=f'best of {reps} calls of matcher.section_matcher on resident uint8 sections; error = distance of the matched displacement '
                     'from the injected field at the matched points; the relaxations between the rounds stop at the reference tolerance '
                     '0.01 / max(1, max_dis) (matcher.py:685-688; relax_tol=1e-9 converges them: 2 x ~1 400 iterations, +40 ms, same error)')


def cpu_baseline_ncc(h0, h1, seconds=20.0):
    """the oracle pair pipeline on the host, one process / one thread, on a bounded sample"""
    from oracle import pipeline_ref
    done = 0
    t0 = time.time()
    while done < h0.shape[0]:
        pipeline_ref.match_pair(h0[done], h1[done], residue_len=RESIDUE_LEN)
        done += 1
        if time.time() - t0 > seconds:
            break
    dt = time.time() - t0
    return done / dt, done, dt


def run_jobs(jobs, ctxs, work, _lib):
    """deal `jobs` round-robin to one host thread per context (thread t takes jobs t, t + T, ...) and return the results
    in job order; every thread drives the device through its own context (HIP stream)"""
    import threading
    T = max(1, len(ctxs))
    out = [None] * len(jobs)
    errs = []

    def worker(t):
        try:
            with _lib.using(ctxs[t]):
                for j in range(t, len(jobs), T):
                    out[j] = work(t, jobs[j])
        except Exception as e:                            # noqa: BLE001 -- re-raised in the caller
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errs:
        raise errs[0]
    return out


def bench_stitch_sections(args, lib, _lib, rank, world, ctxs, ex, barrier, reduce_max):
    """config[3] at weak scaling: every rank owns --stitch-sections whole sections (the contiguous slices of the section
    list, stitch_main.py:146-159 / stitcher.py:375-392) of 20 x 20 tiles: 380 left-right + 380 up-down edge overlaps
    (strips 4096 x 510 / 510 x 4096) and 722 diagonal corner overlaps (510 x 510) per section, all resident in HBM.
    Edge pairs and corner pairs are timed separately; then ONE gather of the whole match table to rank 0."""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    from feabas_amd import dist as fdist
    nsec = args.stitch_sections
    T = args.tile
    ov = int(round(0.1 * T)) + 100
    ov += ov % 2
    P = args.sub_batch
    kinds = {'LR': (T, ov, 380 * nsec), 'UD': (ov, T, 380 * nsec), 'corner': (ov, ov, 722 * nsec)}
    strips, jobs = {}, {'edge': [], 'corner': []}
    sec0 = rank * nsec                                     # first global section of this rank
    base = 0
    with _lib.using(ctxs[0]):
        for ki, (k, (H, W, n)) in enumerate(kinds.items()):
            s0 = _lib.DeviceBuffer(n * H * W); s1 = _lib.DeviceBuffer(n * H * W); sh = _lib.DeviceBuffer(n * 8)
            done = 0
            while done < n:                                # seeds: unique per (kind, global section, pair)
                m = min(4096, n - done)
                _lib.check(lib.fb_synth_strips_dev(ctxs[0], m, 10000000 * (ki + 1) + sec0 * 1000 + done, H, W, 2027, 20, 1, args.warp,
                                                   s0.offset(done * H * W), s1.offset(done * H * W), sh.offset(done * 8)))
                done += m
            strips[k] = (s0, s1, sh.to_array((n, 2), np.int32), base)
            base += n
        _lib.check(lib.fb_sync(ctxs[0]))
    for k in ('LR', 'UD'):                                 # interleaved so that both orientations are in flight
        n = kinds[k][2]
        jobs['edge'] += [(k, a, min(P, n - a)) for a in range(0, n, P)]
    jobs['edge'].sort(key=lambda j: (j[1], j[0]))
    n = kinds['corner'][2]
    jobs['corner'] = [('corner', a, min(P, n - a)) for a in range(0, n, P)]
    matchers = {}

    def work(t, job):
        k, a, cnt = job
        H, W, _ = kinds[k]
        key = (t, k, cnt)
        if key not in matchers:
            matchers[key] = StripBatchMatcher(cnt, H, W, residue_len=RESIDUE_LEN)
        s0, s1, sh, gbase = strips[k]
        res = matchers[key].match(s0.offset(a * H * W), s1.offset(a * H * W))
        pid = res['pair'] + (gbase + a)
        tab = np.concatenate((pid[:, None].astype(np.float32), res['xy0'].astype(np.float32), res['xy1'].astype(np.float32),
                              res['weight'][:, None].astype(np.float32)), axis=1)
        d = res['xy1'] - res['xy0'] + sh[a:a + cnt][res['pair']]
        e = np.abs(d).max(axis=1) if d.size else np.zeros(0)
        return tab, int(res['valid'].sum()), int(np.sum(e < 0.5)), e.astype(np.float32)

    # set-up pass (untimed): every (thread, shape, count) matcher builds its buffers and relaxation system once
    nthr = len(ctxs)
    for grp in ('edge', 'corner'):
        seen, warm = set(), []
        for t in range(nthr):
            for j in range(t, len(jobs[grp]), nthr):
                key = (t, jobs[grp][j][0], jobs[grp][j][2])
                if key not in seen:
                    seen.add(key); warm.append((t, jobs[grp][j]))
        import threading

        def warm_thread(t):
            with _lib.using(ctxs[t]):
                for tt, jb in warm:
                    if tt == t:
                        work(t, jb)
        ths = [threading.Thread(target=warm_thread, args=(t,)) for t in range(nthr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    out = dict(sections_per_rank=nsec, sections=nsec * world, tiles_per_section=400, pairs_per_matcher_call=P)
    tables = []
    for grp in ('edge', 'corner'):
        barrier()
        t0 = time.time()
        res = run_jobs(jobs[grp], ctxs, work, _lib)
        for h in ctxs:
            _lib.check(lib.fb_sync(h), h=h)
        dt_rank = time.time() - t0
        barrier()
        dt = reduce_max(time.time() - t0)
        npairs = sum(j[2] for j in jobs[grp])
        tables += [r[0] for r in res]
        rows = sum(r[0].shape[0] for r in res)
        out[grp] = dict(pairs=npairs * world, pairs_per_s=npairs * world / dt, seconds=dt, pairs_per_s_this_rank=npairs / dt_rank,
                        matched=sum(r[1] for r in res), mean_matches_per_pair=rows / max(npairs, 1),
                        matches_within_half_px_of_truth=sum(r[2] for r in res) / max(rows, 1),
                        # distance to the generator's INTEGER offset: its 0.4 px smooth warp is not subtracted, so the median sits
                        # near 0.25 px and the tail at warp + the +-0.5 clip of the sub-pixel fit (tests/test_gpu_fullsize.py::
                        # test_corner_pairs_batch_vs_oracle: the oracle gives the same distances match by match)
                        distance_to_integer_offset_px_q50_q99_max=[float(v) for v in np.quantile(np.concatenate([r[3] for r in res]), [0.5, 0.99, 1.0])] if rows else None,
                        strip=list(kinds['LR'][:2]) if grp == 'edge' else list(kinds['corner'][:2]))
    table = np.concatenate(tables, axis=0) if tables else np.zeros((0, 6), np.float32)
    if ex is not None:
        dt, rows = gather_table_timed(ex, table, barrier, reduce_max)
        out['gather'] = dict(seconds=dt, rows_on_root=rows, bytes_on_root=None if rows is None else rows * 24, backend=ex.backend,
                             note='ONE gather of the float32 match table (pair id, xy0, xy1, weight: 24 B per match, the record of stitcher.py:144-151) '
                                  'to rank 0: counts, then point-to-point transfers of exactly the bytes each rank holds')
        out['edge']['pairs_per_s_incl_gather'] = out['edge']['pairs'] / (out['edge']['seconds'] + dt)
    else:
        out['gather'] = dict(seconds=0.0, rows_on_root=int(table.shape[0]), bytes_on_root=int(table.nbytes), backend='none (one rank)')
    for m in matchers.values():
        m.free()
    for s0, s1, _, _ in strips.values():
        s0.free(); s1.free()
    return out


def bench_align_sections(args, lib, ctx, _lib, rank, world, ex, barrier, reduce_max):
    """config[4] at weak scaling: every rank owns --align-sections sections (contiguous slice of the section list) of
    grid x grid nodes; section g is linked to g - 1 and g + 1 by 50 k matches each and relaxed against them as LOCKED
    neighbours through SLM.optimize_linear (the independent-unit mode of SURVEY.md sec.8e; aligner.py:696-727 with one
    free section), then the node displacements of all sections are all-gathered once."""
    from feabas_amd import mesh, optimizer
    n = args.align_grid
    nsec = args.align_sections
    h = 20.0
    xs = h * np.arange(n)
    vx, vy = np.meshgrid(xs, xs)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tri = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1))).astype(np.int32)
    L = h * (n - 1)

    def field(g):                                           # smooth distortion of section g (a function of g alone: no exchange needed)
        ph = 1.1 * g
        return np.stack((6 * np.sin(2 * np.pi * v[:, 1] / L + ph) + 2 * np.cos(4 * np.pi * v[:, 0] / L - ph),
                         5 * np.cos(2 * np.pi * v[:, 0] / L - ph) + 2 * np.sin(4 * np.pi * v[:, 1] / L + ph)), axis=-1)
    # sections are independent units: T host threads, each with its own context (HIP stream), its own three meshes and SLM
    # (one symbolic pattern per thread, kept across its sections), take the sections round-robin -- the host part of a section
    # (link terms, set_field, the Python around them) of one thread runs beside the device part of another
    T = max(1, min(args.align_threads, nsec, _lib.cpu_budget()))
    actxs = [ctx] + [_lib.new_context() for _ in range(T - 1)]
    units = []
    for t_ in range(T):
        prev = mesh.Mesh(v.copy(), tri, uid=0, locked=True)
        cur = mesh.Mesh(v.copy(), tri, uid=1)
        nxt = mesh.Mesh(v.copy(), tri, uid=2, locked=True)
        units.append((prev, cur, nxt, optimizer.SLM([prev, cur, nxt], [], stiffness_lambda=1.0, crosslink_lambda=-1.0)))
    nl = 50000
    g0 = 1 + rank * nsec                                    # section 0 of the stack is the locked anchor
    disp = np.empty((nsec, n * n, 2))
    from feabas_amd import constant as const
    # the matches of every section pair (what the matching stage delivers, in the raster order of its block grid) and the
    # neighbours' fields are inputs: made before the clock starts
    inputs = []
    for k in range(nsec):
        g = g0 + k
        rng = np.random.default_rng(7000 + g)
        dg = field(g)
        lk = [(np.sort(rng.integers(0, tri.shape[0], nl)), rng.dirichlet((1, 1, 1), nl), rng.uniform(0.3, 1.0, nl).astype(np.float32)) for _ in range(2)]
        inputs.append((v + (field(g - 1) - dg), v + (field(g + 1) - dg), lk))
    zero = np.zeros((1, 2))

    def one_section(t_, k):
        prev, cur, nxt, slm = units[t_]
        vp, vn, lk = inputs[k]
        for m_, vv in ((prev, vp), (nxt, vn)):
            m_.unlock(); m_.set_vertices(vv, const.MESH_GEAR_MOVING); m_.lock()
        cur.set_vertices(v.copy(), const.MESH_GEAR_MOVING); cur.set_offset(zero, const.MESH_GEAR_MOVING)
        slm.links = [optimizer.Link(m0, m1, tid, tid, B, B, weight=w) for (m0, m1), (tid, B, w) in zip(((prev, cur), (cur, nxt)), lk)]
        t1 = time.time()
        slm.optimize_linear(tol=1e-4)
        ts = time.time() - t1
        disp[k] = cur.vertices_w_offset(const.MESH_GEAR_MOVING) - v
        return slm.last_solve['iters'], slm.last_solve['relres'], ts
    barrier()
    t0 = time.time()
    res = run_jobs(list(range(nsec)), actxs, one_section, _lib)
    iters = int(sum(r[0] for r in res)); relres = [r[1] for r in res]; t_solve = float(sum(r[2] for r in res))
    slm = units[(nsec - 1) % T][3]                          # the unit that solved the last section (checked below)
    dt_rank = time.time() - t0
    barrier()
    dt = reduce_max(time.time() - t0)
    # check of the last section (outside the clock): the TRUE residual of its system, recomputed on the host with scipy from the
    # A and b the device assembled and the field the Mesh holds afterwards.  (Round 2 compared the field with the unweighted
    # mean of the neighbours' fields; the two link sets sit at different points and pull a soft mesh towards two different
    # fields, so that distance -- 1.13 -- validated nothing.  Parity of this unit against the reference is golden G18,
    # tests/test_gpu_fem.py::test_g18_section_between_locked_neighbours_vs_reference.)
    from feabas_amd.mesh import bsr_download
    k = nsec - 1
    with _lib.using(actxs[(nsec - 1) % T]):
        A_h = bsr_download(slm._sys, 4, slm._nv, slm._nnzb)
        b_h = np.empty(2 * slm._nv)
        _lib.check(lib.fb_sys_get(_lib.ctx(), slm._sys, 5, _lib.ptr(b_h)))
    d_h = disp[k].ravel()                                  # vertices + offset - start = the solved displacement (set_field, mesh.py:2400-2413)
    true_relres = float(np.linalg.norm(A_h @ d_h - b_h) / np.linalg.norm(b_h))
    out = dict(sections_per_rank=nsec, sections=nsec * world, nodes_per_section=n * n, dof_per_section=2 * n * n, links_per_section=2 * nl,
               sections_per_s=nsec * world / dt, seconds=dt, seconds_this_rank=dt_rank, pcg_iters_this_rank=iters, optimize_linear_seconds_this_rank=t_solve,
               worst_relres=float(max(relres)), true_relres_last_section_recomputed_on_host=true_relres, host_threads=T,
               optimize_linear_s_first_of_a_thread_and_median=[float(np.max([r[2] for r in res[:T]])), float(np.median([r[2] for r in res]))],
               note='per section: link set-up (host), device assembly, Jacobi-PCG to 1e-4 through SLM.optimize_linear; the symbolic pattern is kept across '
                    'sections (matches against locked neighbours stay inside the triangles of the free mesh: fb_sys_update_links); '
                    f'{T} host threads with a context each take the sections round-robin; optimize_linear_seconds_this_rank sums the threads')
    if ex is not None:
        dtg, allx = allgather_timed(ex, disp, barrier, reduce_max)
        out['allgather'] = dict(seconds=dtg, bytes=int(allx.nbytes), backend=ex.backend, shape=list(allx.shape))
        out['sections_per_s_incl_allgather'] = nsec * world / (dt + dtg)
    del slm, units
    for h_ in actxs[1:]:
        lib.fb_destroy(h_)
    return out


def _cpu_pool_worker(task):
    import os as _os
    for var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        _os.environ[var] = '1'
    from oracle import pipeline_ref
    a, b = task
    pipeline_ref.match_pair(a, b, residue_len=RESIDUE_LEN)
    return 1


def cpu_baseline_pool(h0, h1, seconds):
    """the reference's own scaling model (stitching_configs.yaml:18, config.py:301-310): a pool of worker processes, one
    pair at a time each, one BLAS thread per worker, workers = the cores this process may run on"""
    import multiprocessing as mp
    # ... and its cgroup quota allows (a GPU box shows 256 logical CPUs to a container with a quota of 16: round 2 started 256
    # workers there and called the result the figure of 256 cores)
    from feabas_amd import _lib as _fl
    cores = _fl.cpu_budget()
    per_pair = 0.3                                         # about 0.27 s per pair on one core
    ntask = int(max(cores, min(h0.shape[0] * 8, cores * seconds / per_pair)))
    tasks = [(h0[k % h0.shape[0]], h1[k % h0.shape[0]]) for k in range(ntask)]
    ctx_mp = mp.get_context('spawn')
    with ctx_mp.Pool(cores) as pool:
        pool.map(_cpu_pool_worker, tasks[:cores])          # imports + first-call set-up outside the timed part
        t0 = time.time()
        pool.map(_cpu_pool_worker, tasks, chunksize=1)
        dt = time.time() - t0
    return ntask / dt, ntask, dt, cores


def fem_cpu_krylov(A, b, seconds=12.0):
    """scipy.sparse.linalg cg and minres with the reference's Jacobi preconditioner (optimizer.py:1962-1966)
    on the identical A, b: iterations/s and time to ||Ax - b|| <= 1e-4 ||b|| (bounded by `seconds` per solver)"""
    from scipy import sparse
    from scipy.sparse import linalg as sla
    dg = A.diagonal()
    cond = dg.max() / 1000.0
    M = sparse.diags(1.0 / np.clip(dg, min(1.0, cond), None))           # optimizer.py:1962-1966
    bn = np.linalg.norm(b)
    out = {}
    for name, fn in (('cg', sla.cg), ('minres', sla.minres)):
        st = dict(it=0, t0=time.time(), hit=None)

        def cb(xk):
            st['it'] += 1
            if st['hit'] is None and st['it'] % 10 == 0:
                if np.linalg.norm(A @ xk - b) <= 1e-4 * bn:
                    st['hit'] = (st['it'], time.time() - st['t0'])
            if time.time() - st['t0'] > seconds or st['hit'] is not None:
                raise StopIteration
        try:
            if name == 'cg':
                fn(A, b, M=M, rtol=1e-12, maxiter=100000, callback=cb)
            else:
                fn(A, b, M=M, rtol=1e-12, maxiter=100000, callback=cb)
        except StopIteration:
            pass
        el = time.time() - st['t0']
        out[name] = dict(iters_per_s=st['it'] / el, iterations=st['it'], seconds=el,
                         to_1e4=None if st['hit'] is None else dict(iterations=st['hit'][0], seconds=st['hit'][1]),
                         note='true residual checked every 10 iterations (its SpMV is inside the timing)')
    return out


def guarded(line, name, fn, *a, **k):
    """a side record (one rank, no collective inside) must never take the headline down with it: an exception becomes the record"""
    try:
        line[name] = fn(*a, **k)
    except Exception as e:                                # noqa: BLE001 -- reported in the line, traceback on stderr
        import traceback
        traceback.print_exc()
        line[name] = dict(error=f'{type(e).__name__}: {e}')


def gather_table_timed(ex, table, barrier, reduce_max):
    """ONE gather of a rank's float32 match table to rank 0 (Exchange.gatherv: counts, then exactly the bytes every rank
    holds), bracketed like every timed region.  Returns (seconds, rows on the root or None off the root)."""
    barrier()
    t0 = time.time()
    parts = ex.gatherv(table, root=0)
    barrier()
    dt = reduce_max(time.time() - t0)
    return dt, (int(sum(p_.shape[0] for p_ in parts)) if parts is not None else None)


def allgather_timed(ex, arr, barrier, reduce_max):
    """ONE all-gather of equal blocks (node displacements), bracketed like every timed region.  Returns (seconds, [world, ...])"""
    barrier()
    t0 = time.time()
    allx = ex.allgather(arr)
    barrier()
    return reduce_max(time.time() - t0), allx


def match_table(batch, P):
    """the float32 match table of a step: (pair id, xy0, xy1, weight) = 24 B per match, the record the reference stores
    (stitcher.py:144-151); batch: [(index of the matcher call inside the run, result dict)]"""
    pid = np.concatenate([res['pair'] + i * P for i, res in batch])
    return np.concatenate((pid[:, None].astype(np.float32), np.concatenate([r['xy0'] for _, r in batch]).astype(np.float32),
                           np.concatenate([r['xy1'] for _, r in batch]).astype(np.float32),
                           np.concatenate([r['weight'] for _, r in batch])[:, None].astype(np.float32)), axis=1)


def run_steps_threaded(idx, step, exchange, S, nthr, ctxs, use_context, stagger_ms):
    """the matcher calls `idx` dealt round-robin to `nthr` host threads (thread k works on context ctxs[k % len(ctxs)]), so that
    one thread's block-list bookkeeping overlaps the others' kernels (the library serialises calls per context; every call runs
    completely inside the timed region).  `step(i)` returns a tuple whose third entry is the result of call i; every S results
    -- in call order, the same order on every rank -- go to `exchange` from ONE extra thread while the workers go on with later
    calls.  Returns the last call's tuple.  (Module level so that `--dry-run` drives the same code over gloo without a GPU.)"""
    idx = list(idx)
    if not idx:
        return None
    if nthr <= 1:
        out = None
        pending = []
        for i in idx:
            out = step(i)
            pending.append((i, out[2]))
            if len(pending) == S:
                exchange(pending); pending = []
        exchange(pending)
        return out
    import threading
    results = {}
    errs = []
    cv = threading.Condition()

    def worker(mine, h, k=0):
        use_context(h)
        try:
            if stagger_ms:
                time.sleep(1e-3 * stagger_ms * k)
            for i in mine:
                if errs:
                    return
                r = step(i)
                with cv:
                    results[i] = r
                    cv.notify_all()
        except Exception as e:                        # noqa: BLE001 -- re-raised below; the other threads must not wait for this one
            with cv:
                errs.append(e)
                cv.notify_all()

    def comm():
        # a failed gather must not pass for a finished one: the error joins `errs`, the workers stop at their next step and the
        # exception leaves main() -- a non-zero exit, on which the launcher tears the other ranks (blocked in their transfer) down
        try:
            pending = []
            for i in idx:
                with cv:
                    cv.wait_for(lambda: i in results or errs)
                    if errs:
                        return
                    r = results[i]
                pending.append((i, r[2]))
                if len(pending) == S:
                    exchange(pending); pending = []
            exchange(pending)
        except BaseException as e:                    # noqa: BLE001
            with cv:
                errs.append(e)
                cv.notify_all()
    T = nthr
    ths = [threading.Thread(target=worker, args=([i for i in idx if i % T == k], ctxs[k % len(ctxs)], k)) for k in range(T)]
    ths.append(threading.Thread(target=comm))
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return results[idx[-1]]


def dry_run(args, rank, world):
    """no GPU: the ranks run the HOST side of every sharded leg of this file over gloo on synthetic results -- the same
    scheduler (run_steps_threaded), table builder and timed exchange helpers the real run calls, with the shapes and dtypes the
    legs hand them -- so that the first run on N GPUs cannot fail on host logic (tests/test_cpu_bench_spawn.py)"""
    import torch
    import torch.distributed as dist
    from feabas_amd import dist as fdist
    ex = fdist.Exchange() if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()

    def reduce_max(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def total(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        return float(tt.item())
    rng = np.random.default_rng(100 + rank)
    out = dict(metric='tile_pair_ncc_matches_per_s', value=None, unit='pairs/s', n_gpus=world, dry_run=True, ranks=world,
               exchange_backend=ex.backend if ex is not None else 'none')
    # ---- sharding of a global pair list + one gather of a ragged table (the pre-round-5 check, kept)
    nsec = max(1, args.stitch_sections)
    pairs = (760 + 722) * nsec * world
    a, b = fdist.shard_range(pairs, rank, world)
    rows = rng.integers(1, 5, b - a)
    tab = np.concatenate([np.full((r, 6), a + i, dtype=np.float32) for i, r in enumerate(rows)])
    parts = ex.gatherv(tab, root=0) if ex is not None else [tab]
    if rank == 0:
        ids = np.concatenate([p_[:, 0] for p_ in parts])
        out.update(pair_shards_cover_the_list=bool(np.all(np.diff(ids) >= 0) and ids[0] == 0 and ids[-1] == pairs - 1 and np.unique(ids).size == pairs),
                   gathered_rows=int(ids.size))
    # ---- headline leg: S matcher calls per step on several host threads, ONE gather per step issued in call order
    P, S, nthr, W, K = 8, 4, 4, 1, 3
    sent = dict(calls=0, rows=0, rows_on_root=0)

    def step(i):                                           # a matcher call: (kind, block, result) with a ragged number of matches
        r_ = np.random.default_rng(1000 * rank + i)
        n = int(r_.integers(P, 6 * P))
        res = dict(pair=np.sort(r_.integers(0, P, n)), xy0=r_.random((n, 2)), xy1=r_.random((n, 2)), weight=r_.random(n))
        time.sleep(0.002 * float(r_.random()))             # calls finish out of order; the gathers must not
        return 'LR', i % 3, res

    def exchange(batch):
        if ex is not None and batch:
            tab_ = match_table(batch, P)
            assert tab_.dtype == np.float32 and tab_.shape[1] == 6
            got = ex.gatherv(tab_, root=0)
            sent['calls'] += 1; sent['rows'] += tab_.shape[0]
            if got is not None:
                sent['rows_on_root'] += int(sum(p_.shape[0] for p_ in got))
    barrier()
    run_steps_threaded(range(0, W * S), step, exchange, S, nthr, [None], lambda h: None, 1.0)
    barrier()
    t0 = time.time()
    last = run_steps_threaded(range(W * S, (W + K) * S), step, exchange, S, nthr, [None], lambda h: None, 1.0)
    barrier()
    dt = reduce_max(time.time() - t0)
    rows_all = total(sent['rows'])
    if rank == 0:
        out['headline'] = dict(gather_calls=sent['calls'], rows_on_root=sent['rows_on_root'], rows_sent_by_all_ranks=int(rows_all), seconds=dt,
                               last_call=int(last[1]), ok=bool(world == 1 or (sent['calls'] == W + K and sent['rows_on_root'] == int(rows_all))))
    # ---- stitch_sections leg: the table of a rank's sections (edge + corner pairs), ONE gather
    tables = [np.asarray(rng.random((int(rng.integers(50, 90)), 6)), dtype=np.float32) for _ in range(2)]
    table = np.concatenate(tables, axis=0)
    if ex is not None:
        dtg, rows_root = gather_table_timed(ex, table, barrier, reduce_max)
    else:
        dtg, rows_root = 0.0, int(table.shape[0])
    rows_all = total(table.shape[0])
    if rank == 0:
        out['stitch_sections'] = dict(rows_on_root=rows_root, seconds=dtg, ok=bool(rows_root == int(rows_all)))
    # ---- align_sections leg: node displacements [sections of the rank][nodes][2], ONE all-gather
    nalign, nn = max(1, args.align_sections), 4
    sec = fdist.shard_range(nalign * world, rank, world)
    disp = np.full((sec[1] - sec[0], nn, 2), float(rank))
    if ex is not None:
        dta, allx = allgather_timed(ex, disp, barrier, reduce_max)
    else:
        dta, allx = 0.0, disp[None]
    out_shape = list(allx.shape)
    ok_align = bool(allx.shape == (world, nalign, nn, 2) and all(np.all(allx[r] == float(r)) for r in range(world)))
    # ---- fem leg: the displacement vector of a rank's system, ONE all-gather, and the sum of the rates
    x = np.full((37, 2), 10.0 + rank)
    if ex is not None:
        dtf, allf = allgather_timed(ex, x.reshape(-1, 2), barrier, reduce_max)
    else:
        dtf, allf = 0.0, x[None]
    rate = total(1000.0 + rank)
    if rank == 0:
        out.update(allgather_shape=out_shape, align_sections=dict(ok=ok_align, seconds=dta),
                   fem=dict(ok=bool(allf.shape == (world, 37, 2) and np.all(allf[:, 0, 0] == 10.0 + np.arange(world))
                                    and rate == sum(1000.0 + r for r in range(world))), seconds=dtf, allgather_bytes=int(allf.nbytes)))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        fdist.release_exchanges()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        spawn_ranks(args)                                  # does not return
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # resident-set watchdog (feabas_amd/_watchdog.py): the bench holds ~10 GB of host tiles for its PCIe-inclusive leg; anything
    # far above that is a runaway host loop and must end this process, not the box.  FEABAS_RSS_LIMIT_GB (0 = off).
    from feabas_amd import _watchdog
    os.environ.setdefault('FEABAS_RSS_LIMIT_GB', '40')
    if _watchdog.start() > 0:
        _watchdog.start_backstop()
    dist = None
    torch = None
    launched = world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ)      # by torch.distributed.run
    if args.dry_run:
        if launched:
            import torch.distributed as dist
            dist.init_process_group('gloo')
        dry_run(args, rank, world)
        return
    rccl_ranks = 1
    if launched:
        # torch.distributed is the launcher's rendezvous, the barrier and the max-over-ranks of the timings: a gloo group on
        # host tensors.  Every byte of the data path goes through the C ABI on the library's own RCCL communicator, so ONE
        # ROCm stack touches the device (torch's wheel bundles a second one; round 2 initialised both).
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo')
    os.environ['FEABAS_HIP_DEVICE'] = str(local_rank)

    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    lib, ctx = _lib.load(), _lib.ctx(local_rank)
    ex = None
    comm_ctx = None
    if dist is not None and (world > 1 or os.environ.get('FEABAS_BENCH_EXCHANGE') == '1'):       # the env switch: a 1-rank rehearsal of the N-rank path
        # the exchange steps go through the C ABI (fb_comm_* / fb_gatherv_dev / fb_allgather_dev) on a context of their own, so
        # that a gather never queues behind the kernels of a matcher thread
        from feabas_amd import dist as fdist
        comm_ctx = _lib.new_context(local_rank)
        ex_note = None
        try:
            ex = fdist.Exchange(backend='rccl', ctx=comm_ctx)
            ok = 1.0
        except Exception as e:                            # noqa: BLE001 -- reported in the line; the run goes on over the host group
            ex_note = f'C-ABI RCCL communicator unavailable ({e}); exchange through the gloo group on host arrays'
            ok = 0.0
        if fdist.host_sum([ok]) [0] < world:              # one rank without it: every rank takes the host route (collectives must match)
            if not args.allow_host_exchange:
                # a run whose exchange steps went over gloo on host arrays must not pass for an RCCL measurement
                if rank == 0:
                    print(f'bench.py: the C-ABI RCCL communicator could not be made on every rank ({ex_note or "another rank failed"}); '
                          'pass --allow-host-exchange to run the exchange steps over the gloo group instead', file=sys.stderr, flush=True)
                sys.exit(3)
            if ex_note is None:
                ex.close()
                ex_note = 'another rank has no C-ABI RCCL communicator; exchange through the gloo group on host arrays'
            ex = fdist.Exchange(backend='torch')
        ex.note = ex_note
        rccl_ranks = ex.rccl_ranks()                      # an all-reduce of ones through fb_allreduce_f64_dev: the ranks of the LIBRARY's communicator

    def reduce_max(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    T = args.tile
    ov = int(round(0.1 * T)) + 100                        # 10 % overlap + margin 100 (stitching_configs.yaml:21)
    ov += ov % 2
    P = args.sub_batch                                     # pairs per matcher call
    S = max(1, args.pairs_per_step // P)                   # matcher calls per step
    n_res = max(P, args.resident_pairs // 2 // P * P)     # resident pairs per orientation
    shapes = {'LR': (T, ov), 'UD': (ov, T)}
    strips = {}
    matchers = {}
    # host threads of this rank: the flag, capped by the CPUs the rank may use (the ranks of a node share its quota; the threads mostly wait
    # for their stream, so a floor of 4 stays -- fewer leaves the device idle between sub-batches)
    budget = _lib.cpu_budget()
    if args.host_threads > max(4, budget):
        args.host_threads = max(4, budget)
    mper = max(1, args.host_threads // 2)          # matcher instances per orientation: one per host thread
    for k, (H, W) in shapes.items():
        s0 = _lib.DeviceBuffer(n_res * H * W); s1 = _lib.DeviceBuffer(n_res * H * W); sh = _lib.DeviceBuffer(n_res * 8)
        _lib.check(lib.fb_synth_strips_dev(ctx, n_res, rank * 100000 + (0 if k == 'LR' else 50000), H, W, 2026, 20, 1, args.warp,
                                           s0.ptr, s1.ptr, sh.ptr))
        strips[k] = (s0, s1, sh.to_array((n_res, 2), np.int32))
        for j in range(mper):
            matchers[(k, j)] = StripBatchMatcher(P, H, W, residue_len=RESIDUE_LEN)
    _lib.check(lib.fb_sync(ctx))

    def step(i):
        k = 'LR' if i % 2 == 0 else 'UD'
        H, W = shapes[k]
        s0, s1, _ = strips[k]
        b = (i // 2) % (strips[k][0].nbytes // (P * H * W))
        res = matchers[(k, (i // 2) % mper)].match(s0.offset(b * P * H * W), s1.offset(b * P * H * W))
        return k, b, res

    gather_stats = dict(calls=0, rows_on_root=0)

    def exchange(batch):
        # the one exchange of the sharded run: every rank's match table of a STEP (S matcher calls) -> rank 0, one gather
        # (counts, then exactly the bytes of every rank over RCCL point-to-point: fb_gatherv_dev)
        if ex is not None and batch:
            parts = ex.gatherv(match_table(batch, P), root=0)
            gather_stats['calls'] += 1
            if parts is not None:
                gather_stats['rows_on_root'] += int(sum(p_.shape[0] for p_ in parts))

    # every host thread drives the device through its own context (stream): a thread's synchronisation then waits for
    # its own kernels only, and kernels of different batches may overlap on the device
    nthr = max(2, args.host_threads // 2 * 2) if args.host_threads > 1 else 1
    ctxs = [ctx] + [_lib.new_context(local_rank) for _ in range(nthr - 1)] if args.multi_stream and nthr > 1 else [ctx]

    def barrier():
        for h in ctxs:
            _lib.check(lib.fb_sync(h))
        if dist is not None:
            dist.barrier()                                # gloo; the device side of the bracket is the fb_sync of every context above
            for h in ctxs:
                _lib.check(lib.fb_sync(h))

    stagger_ms = float(os.environ.get("FEABAS_HIP_STAGGER_MS", os.environ.get("FEABAS_BENCH_STAGGER_MS", "3")))      # first call of host thread k offset by k x this: the stagger matcher.stitching_matcher_batch applies to its worker threads

    def run_steps(first, count):
        return run_steps_threaded(list(range(first, first + count)), step, exchange, S, nthr if args.host_threads > 1 else 1, ctxs,
                                  _lib.use_context, stagger_ms)

    # one-time set-up outside the step count: every matcher instance builds its resident relaxation system, twiddle
    # tables and scratch arena on its first batch (lazy), and every context allocates its arena on first use -- so each
    # host thread runs its own matcher once on its own context before the W warm-up steps
    def setup_thread(k):
        if len(ctxs) > 1:
            _lib.use_context(ctxs[k % len(ctxs)])
        step(k)
    if nthr > 1:
        import threading
        sts = [threading.Thread(target=setup_thread, args=(k,)) for k in range(nthr)]
        for t in sts:
            t.start()
        for t in sts:
            t.join()
    else:
        step(0); step(1)
    barrier()
    run_steps(0, args.warmup * S)
    barrier()
    for h in ctxs:
        _lib.check(lib.fb_prof_reset(h)); _lib.check(lib.fb_prof_enable(h, 1))
    t0 = time.time()
    last = run_steps(args.warmup * S, args.steps * S)
    barrier()
    dt = time.time() - t0
    prof_timed = {}
    for h in ctxs:
        _lib.check(lib.fb_prof_enable(h, 0))
        for k_, v in _lib.prof_snapshot(h).items():
            o = prof_timed.get(k_, (0, 0.0, 0.0))
            prof_timed[k_] = (o[0] + v[0], o[1] + v[1], o[2] + v[2])
    prof = prof_timed
    iso_steps = 0
    if len(ctxs) > 1:
        # With one stream per host thread the kernels of different batches overlap on the device, so the event durations
        # taken inside the timed region are not per-kernel costs.  The roofline numbers come from a few more steps of the
        # same workload issued on ONE stream (nothing else on the device), after the timed region.
        iso_steps = 4
        _lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
        for i in range(iso_steps):
            step((args.warmup + args.steps) * S + i)
        _lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_enable(ctx, 0))
        prof = _lib.prof_snapshot(ctx)
    dt_rank = dt
    dt = reduce_max(dt)
    pairs = args.steps * S * P * world
    per_rank = [args.steps * S * P / dt_rank]
    if dist is not None:
        tl = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([per_rank[0]], dtype=torch.float64))
        per_rank = [float(t_.item()) for t_ in tl]

    # correctness of the timed work: recovered translations = injected shifts, matches found
    k, b, res = last
    sh = strips[k][2][b * P:(b + 1) * P]
    ok_shift = int(np.sum((np.abs(res['tx'] + sh[:, 0]) <= 1) & (np.abs(res['ty'] + sh[:, 1]) <= 1)))
    d = res['xy1'] - res['xy0'] + sh[res['pair']]
    ok_match = float(np.mean(np.abs(d).max(axis=1) < 0.5)) if d.size else 0.0
    n_matches = res['pair'].size / P

    # roofline of the dominant kernel (by accumulated event time inside the timed region)
    dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else (None, (0, 0.0, 0.0))
    Hl, Wl = shapes['LR']
    fine_w = -(-Wl // 7)
    pair_bytes = ncc_bytes_model(Hl, Wl, [(4, Hl // 4, Wl, True), (385, 75, fine_w, False)])
    total_ms = sum(v[1] for v in prof.values())
    roof = dict(bound='hbm', kernel=dom[0], launches=dom[1][0],
                avg_launch_ms=(dom[1][1] / max(dom[1][0], 1)), share_of_gpu_time=(dom[1][1] / max(total_ms, 1e-9)),
                achieved=None, peak=HBM_PEAK_GBS, unit='GB/s', frac=None, traffic=None,
                pipeline_algorithmic_bytes_per_pair=pair_bytes,
                pipeline_achieved_gbs=pair_bytes * pairs / world / dt / 1e9,
                pipeline_frac=pair_bytes * pairs / world / dt / 1e9 / HBM_PEAK_GBS,
                kernel_ms={k_: round(v[1], 3) for k_, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
                kernel_ms_steps=(iso_steps if iso_steps else args.steps * S),       # matcher calls the kernel_ms totals cover
                avg_launch_ms_timed_region=(prof_timed[dom[0]][1] / max(prof_timed[dom[0]][0], 1) if dom[0] in prof_timed else None),
                measured_on=(f'{iso_steps} extra matcher calls ({P} pairs each) of the same workload on one stream after the timed region (the timed region runs one '
                             f'stream per host thread: its kernels overlap, event sum {sum(v[1] for v in prof_timed.values()) / max(args.steps * S, 1):.2f} '
                             f'ms per sub-batch against {1e3 * dt / max(args.steps * S, 1):.2f} ms wall -- avg_launch_ms_timed_region is that overlapped duration and is what '
                             f'rocprofv3 --stats of this command reports, profiles/*_bench_kernel_stats.csv); --multi-stream 0 times the kernels inside '
                             f'the timed region (profiles/*_bench_one_stream_kernel_stats.csv)'
                             if iso_steps else 'the timed region (one stream)'))
    if dom[1][0] > 0 and dom[1][2] > 0:
        # algorithmic bytes of the kernel's launches (accounted by the library from the launch shapes)
        # over their summed duration (HIP events on the context stream around every launch)
        roof['achieved'] = dom[1][2] / (dom[1][1] * 1e-3) / 1e9
        roof['frac'] = roof['achieved'] / HBM_PEAK_GBS
        roof['algorithmic_bytes_per_launch'] = dom[1][2] / dom[1][0]
    # HBM bytes per launch of that kernel from the PMC passes of this same command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    # separate rocprofv3 --pmc runs, tools/pmc_traffic.py); counters cannot be read from inside the timed run
    tfile = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', '*pmc_traffic.json')))
    if tfile and dom[0] is not None:
        tk = json.load(open(tfile[-1]))['kernels']
        # profile label of the library -> device symbols as rocprofv3 names them (the column pass runs in its direct form
        # ncc_cols2_p2 where the correlation is zero padded, in the staged form ncc_cols_p2 elsewhere)
        alias = {'ncc_stream_cols': ('ncc_cols2_p2', 'ncc_cols_p2'), 'ncc_stream_rows': ('ncc_rows_p2',), 'ncc_stream_inv': ('ncc_inv_p2',),
                 'dog_fast': ('dog_stream',)}
        syms = alias.get(dom[0], (dom[0],))
        # the file is keyed by template instance (ncc_cols2_p2<2048>, ...) and comes from a run of THIS command restricted to the
        # timed steps (--no-xcorr-classes ...): the instances are those of the timed shapes, in the mix of the timed steps
        hit = [v for k_, v in tk.items() if k_.split('<')[0] in syms]
        if hit:
            nl = sum(v['launches'] for v in hit)
            roof['traffic'] = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in hit) / max(nl, 1)
            roof['traffic_source'] = 'profiles/' + os.path.basename(tfile[-1])
            roof['kernel_symbols'] = {k_: dict(launches=v['launches'], hbm_bytes_per_launch=v['hbm_bytes_per_launch']) for k_, v in tk.items() if k_.split('<')[0] in syms}
    roof['per_kernel_gbs'] = {k_: round(v[2] / (v[1] * 1e-3) / 1e9, 1) for k_, v in prof.items() if v[1] > 0 and v[2] > 0}
    if rank == 0:
        # what a plain streaming kernel reaches on THIS device at the read : write mixes of the NCC passes (fb_hbm_probe, 1 GiB per
        # stream): `peak` above is the data-sheet read figure the contract asks for; the column and row passes write twice what they
        # read, the inverse pass only reads
        cal = {}
        for name, (nr, nw) in (('read', (1, 0)), ('copy_1r_1w', (1, 1)), ('1r_2w', (1, 2)), ('write', (0, 1))):
            gbs = C.c_double()
            if lib.fb_hbm_probe(ctx, nr, nw, C.byref(gbs)) == 0:
                cal[name] = round(gbs.value, 1)
        if cal:
            mix = {'ncc_stream_cols': '1r_2w', 'ncc_stream_rows': '1r_2w', 'ncc_stream_inv': 'read', 'dog_fast': 'write', 'ncc_small_fused': 'read'}
            roof['hbm_calibration_gbs'] = cal
            roof['hbm_calibration_note'] = ('plain streaming kernels (one contiguous run per workgroup: the form with the best store rate, tools/hbm_store_probe.hip), 1 GiB per stream, measured in this run; frac_of_calibrated = achieved / the rate '
                                            'of the mix the dominant kernel has (ncc_cols: reads 16 S_h, writes 16 S = 1 : 2)')
            if dom[0] in mix and mix[dom[0]] in cal and roof.get('achieved'):
                roof['frac_of_calibrated'] = roof['achieved'] / cal[mix[dom[0]]]
            roof['per_kernel_frac_of_calibrated'] = {k_: round(v / cal[mix[k_]], 3) for k_, v in roof['per_kernel_gbs'].items() if k_ in mix and mix[k_] in cal}

    line = dict(metric='tile_pair_ncc_matches_per_s', value=pairs / dt, unit='pairs/s', n_gpus=world, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling='weak',
                vs_baseline=None, dtype='f32', data='synthetic',
                config=dict(workload=f'config[1]: {2 * n_res} resident synthetic {T}x{T} tile pairs ({n_res} LR + {n_res} UD strips '
                                     f'{Hl}x{Wl}), {P * S} pairs per step ({S} matcher calls of {P} pairs dealt to {nthr} host threads); stages: x0.5 downsample, DoG, global NCC, DoG, '
                                     f'4 coarse + 385 fine block NCCs, last-round relaxation + residue weights + strain (integer synthetic offsets in +-20 px plus a smooth '
                                     f'{args.warp} px warp; odd offsets take the rigid mesh-relaxation branch, pairs whose coarse blocks disagree the deformed-mesh branch, DESIGN.md sec.5)',
                            pairs_per_step=P * S, pairs_per_matcher_call=P, strip=[Hl, Wl], sigma=2.5, conf_thresh=0.33, residue_mode='huber', residue_len=RESIDUE_LEN),
                check=dict(global_shift_within_1px=f'{ok_shift}/{P}', mean_matches_per_pair=n_matches, matches_within_half_px_of_truth=ok_match,
                           pairs_with_deformed_mesh=int(res['deformed'].sum())),
                roofline=roof, rccl_ranks=rccl_ranks, pairs_per_s_per_rank=per_rank,
                match_table_gather=dict(per_step=1, calls=gather_stats['calls'], rows_on_root=gather_stats['rows_on_root'],
                                        seconds_inside_timed_region_and_warmup=(ex.seconds if ex is not None else 0.0),
                                        backend=(ex.backend if ex is not None else 'none (one rank)'), note=(getattr(ex, 'note', None) if ex is not None else None)))

    if world == 1 and not args.no_deformed:
        # the non-rigid branch, driver-visible: the same step on pairs whose strips differ by a smooth 2 px warp, so that the
        # coarse blocks of a pair disagree and its mesh1 is relaxed to a deformed state between the spacings
        n_def = 4 * P
        dstrips = {}
        for k, (H, W) in shapes.items():
            s0 = _lib.DeviceBuffer(n_def * H * W); s1 = _lib.DeviceBuffer(n_def * H * W); sh = _lib.DeviceBuffer(n_def * 8)
            _lib.check(lib.fb_synth_strips_dev(ctx, n_def, 700000 + (0 if k == 'LR' else 50000), H, W, 2026, 20, 1, 2.0, s0.ptr, s1.ptr, sh.ptr))
            dstrips[k] = (s0, s1, sh.to_array((n_def, 2), np.int32))
        _lib.check(lib.fb_sync(ctx))
        keep = dict(strips)
        strips.update(dstrips)
        try:
            w0 = max(S, 2 * nthr)                          # untimed: every host thread's matcher sizes the scratch of the deformed
            run_steps(0, w0)                               # path and learns that its pairs take the general route
            barrier()
            t0 = time.time()
            lastd = run_steps(w0, 4 * S)
            barrier()
            dtd = time.time() - t0
            kd, bd, resd = lastd
            shd = strips[kd][2][bd * P:(bd + 1) * P]
            dd = resd['xy1'] - resd['xy0'] + shd[resd['pair']]
            line['deformed'] = dict(value=4 * S * P / dtd, unit='pairs/s', steps=4, warp_px=2.0, pairs_with_deformed_mesh_last_call=int(resd['deformed'].sum()),
                                    pairs_per_call=P, matches_within_2p5_px_of_rigid_truth=float(np.mean(np.abs(dd).max(axis=1) < 2.5)) if dd.size else 0.0,
                                    mean_matches_per_pair=resd['pair'].size / P)
        finally:
            strips.update(keep)
            for s0, s1, _ in dstrips.values():
                s0.free(); s1.free()

    if not args.no_fem:
        # FEM path (config[2]): every rank relaxes its own ~1e6-DoF section system -- sections are independent SLMs
        # (SURVEY.md sec.8e), no collective on the data path; the node displacements are all-gathered afterwards
        for m in matchers.values():
            m.free()
        args._solo = (world == 1)
        fem = bench_fem(args, lib, ctx, _lib)
        if dist is not None and ex is not None:           # (a one-rank launch under torchrun has a group but nothing to exchange)
            # one all-gather of the node displacements (fb_allgather_dev)
            fem['allgather_displacements_s'], allx = allgather_timed(ex, fem.pop('x').reshape(-1, 2), barrier, reduce_max)
            fem['allgather_bytes'] = int(allx.nbytes)
            tt = torch.tensor([fem['iters_per_s']], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            fem['iters_per_s_all_ranks'] = float(tt.item())
        fem.pop('x', None)
        if rank == 0:
            line['fem'] = fem
    if rank == 0 and not args.no_xcorr_classes:
        guarded(line, 'xcorr_fft_classes', bench_xcorr_classes, lib, ctx, _lib)
    if rank == 0 and world == 1 and args.host_ingest_pairs > 0:
        # the boundary as stitcher.py uses it: strips in host memory.  PCIe-inclusive rate through stitching_matcher_batch
        # (page-locked staging, copy and kernels of different chunks overlapped); never `value`
        from feabas_amd import matcher as fmatcher
        H, W = shapes['LR']
        s0, s1, _ = strips['LR']
        nh = min(args.host_ingest_pairs, n_res)
        IB = max(1, args.host_ingest_batch)
        h0 = s0.to_array((nh, H, W), np.uint8); h1 = s1.to_array((nh, H, W), np.uint8)
        host_pairs = [(h0[k % nh], h1[k % nh]) for k in range(args.host_ingest_pairs)]     # the list may wrap around the resident strips
        cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=RESIDUE_LEN)
        fmatcher.stitching_matcher_batch(host_pairs[:IB * args.host_ingest_threads], batch=IB, threads=args.host_ingest_threads, **cfg)        # set-up pass
        dth = np.inf
        for _ in range(2):                                  # best of two passes: a pass is ~0.12 s and shares the host and the PCIe link with whatever else runs there
            t0 = time.time()
            outp = fmatcher.stitching_matcher_batch(host_pairs, batch=IB, threads=args.host_ingest_threads, **cfg)
            dth = min(dth, time.time() - t0)
        line['host_ingest'] = dict(value=len(host_pairs) / dth, unit='pairs/s', pairs=len(host_pairs), matched=int(sum(o[0] is not None for o in outp)),
                                   h2d_gbs=len(host_pairs) * 2 * H * W / dth / 1e9,
                                   note='strips handed over in host memory (4.2 MB per pair over PCIe), results returned per pair; '
                                        f'{args.host_ingest_threads} host threads ({3 * args.host_ingest_threads // 8} loaders that pack and copy, the others match), {IB}-pair chunks; '
                                        'h2d_gbs = strip bytes over the wall time of the whole run; best of two passes')
        # the same measurement over a list three times as long (two sections' worth of pairs): what the fixed part of a call -- worker
        # start-up, the first chunk's pack + copy before any kernel runs, the last chunk's results -- costs the 0.12 s pass above
        try:
            long_list = host_pairs * 3
            t0 = time.time()
            outl = fmatcher.stitching_matcher_batch(long_list, batch=IB, threads=args.host_ingest_threads, **cfg)
            dtl = time.time() - t0
            line['host_ingest']['three_times_the_list'] = dict(value=len(long_list) / dtl, unit='pairs/s', pairs=len(long_list), seconds=dtl,
                                                               h2d_gbs=len(long_list) * 2 * H * W / dtl / 1e9, matched=int(sum(o[0] is not None for o in outl)),
                                                               fixed_seconds_per_call=max(0.0, (3 * dth - dtl) / 2),
                                                               note='one pass; fixed_seconds_per_call = (3 t(list) - t(3 x list)) / 2')
            del long_list, outl
        except Exception as e:                              # noqa: BLE001 -- a side record
            line['host_ingest']['three_times_the_list'] = dict(error=f'{type(e).__name__}: {e}')
        # the same pairs cropped to strip shapes that all differ (what stage jitter does to the overlaps of a real section):
        # batches of unequal strips (RaggedStripBatchMatcher)
        rng_r = np.random.default_rng(5)
        ragged = []
        for k in range(len(host_pairs)):                   # as many pairs as the uniform list above
            dh, dw = int(rng_r.integers(0, 30)), int(rng_r.integers(0, 12))
            ragged.append((h0[k % nh, :H - dh, :W - dw], h1[k % nh, :H - dh, :W - dw]))
        RB = 32                                             # (48-, 64- and 96-pair chunks are 5-10 % slower; 16 host threads are no faster than 8:
        RT = args.host_ingest_threads                       # the six mesh-grid buckets of this list end in small chunks, gpurun_out sweep of round 4)
        fmatcher.stitching_matcher_batch(ragged, batch=RB, threads=RT, **cfg)      # first pass: page-locked staging, systems, code objects (1.6 s once per process)
        dtr = np.inf
        for _ in range(2):
            t0 = time.time()
            outr = fmatcher.stitching_matcher_batch(ragged, batch=RB, threads=RT, **cfg)
            dtr = min(dtr, time.time() - t0)
        line['host_ingest']['ragged'] = dict(value=len(ragged) / dtr, unit='pairs/s', pairs=len(ragged), distinct_shapes=len({a.shape for a, _ in ragged}),
                                             matched=int(sum(o[0] is not None for o in outr)),
                                             note=f'every pair cropped to its own strip size (up to 29 x 11 px smaller); {RB}-pair chunks of unequal strips dealt to {RT} host threads; best of two passes after a set-up pass')
        # the ragged list again with masks on every 8th pair (tiles with an artefact: stitcher.py:561-571 hands their masks over): since
        # round 6 masked pairs ride in the ragged chunks of their neighbours instead of in per-shape chunks of their own
        try:
            masked = []
            for k, (a, b) in enumerate(ragged):
                if k % 8 == 0:
                    mk = np.ones(a.shape, dtype=bool)
                    mk[: a.shape[0] // 16, : a.shape[1] // 3] = False
                    masked.append((a, b, mk, mk.copy()))
                else:
                    masked.append((a, b))
            fmatcher.stitching_matcher_batch(masked, batch=RB, threads=RT, **cfg)
            dtm = np.inf
            for _ in range(2):
                t0 = time.time()
                outm = fmatcher.stitching_matcher_batch(masked, batch=RB, threads=RT, **cfg)
                dtm = min(dtm, time.time() - t0)
            line['host_ingest']['ragged']['with_masks'] = dict(value=len(masked) / dtm, unit='pairs/s', pairs=len(masked), masked_pairs=len(masked[::8]),
                                                               matched=int(sum(o[0] is not None for o in outm)),
                                                               note='the ragged list with masks on every 8th pair (both strips): masked pairs go through the ragged chunks of their neighbours')
            del masked, outm
        except Exception as e:                              # noqa: BLE001 -- a side record
            line['host_ingest']['ragged']['with_masks'] = dict(error=f'{type(e).__name__}: {e}')
        fmatcher.stitching_matcher_batch_release()
        del h0, h1, host_pairs, outp, ragged, outr
    if rank == 0 and world == 1 and not args.no_align:
        guarded(line, 'align_block_matcher', bench_align, lib, ctx, _lib)
        guarded(line, 'section_matcher', bench_section_matcher, lib, ctx, _lib)
        # the same pair with the matcher_config of the reference's default alignment configuration as it stands
        # (configs/default_alignment_configs.yaml:14-28): spacings [400, 100] x shrink_factor 0.7, sigma 3.5, conf_thresh 0.35,
        # min_boundary_distance 20, residue_len -2 section thicknesses, batch_size 100
        guarded(line, 'section_matcher_default_config', bench_section_matcher, lib, ctx, _lib, reps=2,
                config=dict(spacings=[400, 100], shrink_factor=0.7, sigma=3.5, conf_thresh=0.35, min_boundary_distance=20, residue_mode='huber',
                            residue_len=-2, batch_size=100, pad=True, stiffness_multiplier_threshold=0.1, render_weight_threshold=0.1))
    cpu_strips = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        H, W = shapes['LR']
        s0, s1, _ = strips['LR']
        ns = min(32, n_res)
        cpu_strips = (s0.to_array((ns, H, W), np.uint8), s1.to_array((ns, H, W), np.uint8))
    for s0, s1, _ in strips.values():
        s0.free(); s1.free()
    for m in matchers.values():
        m.free()
    if args.stitch_sections > 0:
        st = bench_stitch_sections(args, lib, _lib, rank, world, ctxs, ex, barrier, reduce_max)
        if dist is not None:
            tl = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(tl, torch.tensor([st['edge'].pop('pairs_per_s_this_rank')], dtype=torch.float64))
            st['edge']['pairs_per_s_per_rank'] = [float(t_.item()) for t_ in tl]
            st['corner'].pop('pairs_per_s_this_rank', None)
        if rank == 0:
            line['stitch_sections'] = st
    if args.align_sections > 0:
        al = bench_align_sections(args, lib, ctx, _lib, rank, world, ex, barrier, reduce_max)
        if dist is not None:
            tt = torch.tensor([float(al['pcg_iters_this_rank']), al['optimize_linear_seconds_this_rank']], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            al['pcg_iters_all_ranks'] = float(tt[0].item())
        if rank == 0:
            line['align_sections'] = al
    if cpu_strips is not None:
        h0, h1 = cpu_strips
        rate, done, secs = cpu_baseline_ncc(h0, h1)
        line['cpu_baseline'] = dict(value=rate, unit='pairs/s', cores=1, kind='port',
                                    sample=f'{done} LR pairs of the same synthetic strips through oracle/pipeline_ref.match_pair '
                                           f'(scipy.fft/ndimage, 1 process, {secs:.1f} s)')
        if args.cpu_pool_seconds > 0:
            # all host cores, the reference's own scaling model: a pool of single-threaded worker processes
            prate, pn, psecs, cores = cpu_baseline_pool(h0, h1, args.cpu_pool_seconds)
            line['cpu_baseline']['all_cores'] = dict(value=prate, unit='pairs/s', cores=cores, kind='port',
                                                     sample=f'{pn} pairs (the same {h0.shape[0]} LR pairs repeated) over a pool of {cores} worker processes, one BLAS thread each, '
                                                            f'{psecs:.1f} s; workers = the CPU quota of this container (affinity and cgroup cpu.max); the host shows {os.cpu_count()} logical CPUs')
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        if ex is not None:
            ex.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
