// Device-resident 2x2-block CSR matrix + PCG workspace (internal).
#pragma once
#include "fb_common.h"

struct fb_bsr_dev {
    int nb = 0;          // block rows (= vertices)
    int* rowptr = nullptr;
    int* col = nullptr;
    double* val = nullptr;   // [nnzb][4] row-major 2x2
    int xcd_rows = 1;        // SpMV: the workgroups of one XCD (blockIdx % 8) take one contiguous eighth of the row chunks
};

struct fb_pcg_state {
    double tol2bb;   // (tol*||b||)^2
    double rr;
    int flag;        // 0 running, 1 converged, 2 breakdown
    int iter;
    double curv_eps; // 1e-9 * max diagonal: the noise level below which p^T A p <= 0 ends the leg instead of failing
};

struct fb_bsr {
    fb_bsr_dev d;
    int64_t nnzb = 0;
    double2 *x = nullptr, *r = nullptr, *z = nullptr, *p0 = nullptr, *p1 = nullptr, *Ap = nullptr, *minv = nullptr, *b = nullptr,
            *diag = nullptr;
    double2* xbest = nullptr;   // the iterate with the best TRUE residual of a solve that needs more than one leg (allocated then)
    double* parts = nullptr;
    fb_pcg_state* state = nullptr;
    double diag_max = 0.0;
    int max_row_blocks = 0;     // longest block row of the pattern (0 = not known): fb_bsr_pcg_groups keeps the matrix of a group in LDS when it fits
    double last_bnorm = 0.0;    // ||b|| of the last fb_bsr_pcg_dev
    // fb_bsr_pcg_dev stops early (probe_stopped) once the iterations it projects from the decay of the residual over the last
    // 64 exceed probe_limit (0: no projection) -- the 'auto' policy of fb_sys_solve hands such a solve to the multigrid
    int probe_limit = 0;
    bool probe_stopped = false;
    // one batch of Jacobi-PCG iterations (no first-iteration special case) as an executable graph: small
    // systems are bound by the launch rate of the two kernels per iteration, not by their run time
    hipGraphExec_t pcg_graph = nullptr;
    int pcg_graph_iters = 0;
    bool pcg_graph_off = false;      // a capture / instantiation failed once: this matrix stays on plain launches
};

struct fb_csr {
    int64_t n = 0, nnz = 0;
    fb_bsr* M = nullptr;
};

int fb_bsr_alloc(fb_ctx* ctx, int nb, int64_t nnzb, fb_bsr** out);
int fb_bsr_free(fb_ctx* ctx, fb_bsr* M);
int fb_bsr_upload(fb_ctx* ctx, int nb, const std::vector<int>& rowptr, const std::vector<int>& col, const std::vector<double>& val,
                  fb_bsr** out);
int fb_bsr_spmv_dev(fb_ctx* ctx, fb_bsr* M, const double2* v, double2* y);
int fb_bsr_setup_jacobi(fb_ctx* ctx, fb_bsr* M, int precond);
int fb_bsr_pcg_dev(fb_ctx* ctx, fb_bsr* M, double rtol, double atol, int maxiter, int fixed_iters, int* iters_out, double* relres_out);
int fb_csr_to_bsr_host(fb_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* idx, const double* val, int symmetrize,
                       int* nb_out, std::vector<int>& browptr, std::vector<int>& bcol, std::vector<double>& bval);

// one workgroup per equal vertex range of a block-diagonal system: whole Jacobi-PCG in LDS (results in M->x)
int fb_bsr_pcg_groups(fb_ctx* ctx, fb_bsr* M, int ngroups, double rtol, double atol, int maxiter, int precond, int* iters_dev, double* relres_dev,
                      int* flags_dev);
