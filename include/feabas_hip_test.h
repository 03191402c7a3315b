/* Test hooks of libfeabas_hip: NOT part of the product library.  They are compiled only into feabas_amd/libfeabas_hip_test.so
 * (the product objects plus the three sources that hold a hook, rebuilt with -DFB_TEST_HOOKS; csrc/Makefile), which the tests
 * and the fuzzers load beside the product library (feabas_amd/_lib.py: load_test()).  libfeabas_hip.so exports only the
 * boundary of include/feabas_hip.h. */
#ifndef FEABAS_HIP_TEST_H
#define FEABAS_HIP_TEST_H
#include "feabas_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test hooks (host only, no context): the host arithmetic fb_match_strips runs between its kernels.  fb_debug_rigid_fits:
 * spatial.fit_affine(xy0, xy1, return_rigid=True, weight, svd_clip=(1, 1)) (matcher.py:752-763) of every pair's matches in one pass --
 * pid [K] ascending runs, p0 / p1 [K][2], wt [K] -> R [P][3][3] (row vectors: p0 ~ p1 @ R), bad [P] = 1 where the pair is rank deficient,
 * reflected or has fewer than 3 matches (the caller's statement-by-statement route).  fb_debug_auto_spacings: matcher.py:243-251 for two
 * strips of H x W, descending.  fb_debug_grid_counts: node counts of Mesh.from_bbox((0, 0, W, H), cartesian=True) (mesh.py:403-435). */
int fb_debug_rigid_fits(int P, int64_t K, const int32_t* pid, const double* p0, const double* p1, const float* wt, double* R, uint8_t* bad);
int fb_debug_auto_spacings(int H, int W, double* out, int cap, int* count);
int fb_debug_grid_counts(int H, int W, double mesh_size, int min_num_blocks, int* nx, int* ny);
/* test hook (host only, no context): one coarsening step of the multigrid set-up (csrc/fb_mg.inc: aggregates = grid cells per mesh,
 * relative node positions, coarse pattern -- the threaded host half of the set-up) on a level given as host arrays: xy [n][2], comp [n]
 * (mesh of every node), the block pattern rowptr [n + 1] / col.  Outputs: *nc aggregates, *cell, agg [n], rel [n][2], and -- when
 * ccol is given with capacity ccol_cap >= *cnnz -- cxy [nc][2], ccomp [nc], crowptr [nc + 1], ccol [*cnnz]; *maxc = longest coarse row. */
int fb_debug_mg_coarsen(int n, int bs, const double* xy, const int32_t* comp, const int32_t* rowptr, const int32_t* col, double fine_scale,
                        int32_t* nc, double* cell, int32_t* agg, double* rel, double* cxy, int32_t* ccomp, int32_t* crowptr, int64_t* cnnz,
                        int32_t* ccol, int64_t ccol_cap, int32_t* maxc);
/* test hook: M complex transforms of length N (5-smooth, <= 4096) through the LDS FFT core the NCC
 * kernels are built on; in/out are host arrays [M][N][2] float32; inverse is un-normalised. */
int fb_debug_fft1d(fb_ctx* ctx, const float* in_host, float* out_host, int M, int N, int inverse, int pad);

#ifdef __cplusplus
}
#endif
#endif
