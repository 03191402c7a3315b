"""Coupled-window solve across ranks (SURVEY.md sec.8e mode 2): a chain of sections, each an elastic-like SPD block, linked to
its neighbours; rows partitioned by section over the ranks; feabas_amd.dist.pcg_row_partitioned with the local rows on the
GPU.  Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/coupled_window.py
(backend nccl = RCCL; --backend gloo stages scalars and halos through the host, e.g. several ranks on one GPU)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def section_rows(sec, nv_side, n_sections, links_per_pair, seed=0):
    """CSR rows (global columns) of section `sec`: a 5-point elastic-like stencil on a nv_side x nv_side grid for both
    coordinates (coupled weakly), plus the link terms w (u_a - u_b)^2 to sections sec - 1 and sec + 1"""
    from scipy import sparse
    nv = nv_side * nv_side
    n = 2 * nv
    lap = sparse.diags([-1.0, -1.0, 4.2, -1.0, -1.0], [-nv_side, -1, 0, 1, nv_side], shape=(nv, nv), format='csr')
    K = sparse.kron(lap, np.array([[1.0, 0.15], [0.15, 1.0]]), format='coo')
    ri, ci, vv = [K.row], [K.col + sec * n], [K.data]
    for other in (sec - 1, sec + 1):
        if other < 0 or other >= n_sections:
            continue
        lo = min(sec, other)
        rng = np.random.default_rng(seed * 1000 + lo)                 # both sides of a boundary draw the same links
        va = rng.integers(0, nv, links_per_pair); vb = rng.integers(0, nv, links_per_pair); w = rng.uniform(0.3, 1.0, links_per_pair)
        mine, theirs = (va, vb) if sec == lo else (vb, va)
        for c in range(2):
            ri += [2 * mine + c, 2 * mine + c]
            ci += [sec * n + 2 * mine + c, other * n + 2 * theirs + c]
            vv += [w, -w]
    return sparse.csr_matrix((np.concatenate(vv), (np.concatenate(ri), np.concatenate(ci))), shape=(n, n * n_sections))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sections-per-rank', type=int, default=2)
    ap.add_argument('--grid', type=int, default=354, help='nodes per side of a section (grid^2 nodes, 2 grid^2 DoF)')
    ap.add_argument('--links', type=int, default=5000)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--rtol', type=float, default=1e-6)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from scipy import sparse
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1')); local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % max(ndev, 1))
    os.environ['FEABAS_HIP_DEVICE'] = str(local % max(ndev, 1))
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    from feabas_amd import dist as fdist
    S = args.sections_per_rank * world
    n = 2 * args.grid * args.grid
    t0 = time.time()
    mine = range(rank * args.sections_per_rank, (rank + 1) * args.sections_per_rank)
    rows = sparse.vstack([section_rows(s, args.grid, S, args.links) for s in mine]).tocsr()
    r0 = mine[0] * n
    part = fdist.RowPartition(rows.indptr, rows.indices, r0)
    dev = torch.device('cuda', torch.cuda.current_device())
    rng = np.random.default_rng(7 + rank)
    b = torch.from_numpy(rng.standard_normal(rows.shape[0])).to(dev)
    diag = np.asarray(rows[np.arange(rows.shape[0]), r0 + np.arange(rows.shape[0])]).ravel()
    minv = torch.from_numpy(1.0 / diag).to(dev)
    spmv = fdist.DeviceRows(part, rows.indptr, rows.data)
    t_setup = time.time() - t0
    dist.barrier(); torch.cuda.synchronize(); t0 = time.time()
    with torch.cuda.stream(spmv.stream()):
        fdist.pcg_row_partitioned(part, spmv, b, minv, rtol=0.5, maxiter=4)         # warm-up (lazy kernel / communicator set-up)
        torch.cuda.current_stream().synchronize(); dist.barrier(); t0 = time.time()
        x, it, rel = fdist.pcg_row_partitioned(part, spmv, b, minv, rtol=args.rtol, maxiter=20000)
        torch.cuda.current_stream().synchronize()
    dist.barrier(); dt = time.time() - t0
    if rank == 0:
        print(f'coupled window: {S} sections x {n} DoF = {S * n} DoF on {world} rank(s), halo {part.n_halo} entries/rank, '
              f'set-up {t_setup:.1f} s; {it} iterations to {rel:.1e} in {dt:.3f} s = {it / dt:.0f} it/s ({1e6 * dt / max(it, 1):.0f} us per iteration)')
    spmv.free()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
