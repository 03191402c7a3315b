/* feabas_hip.h -- C ABI of libfeabas_hip.so, the MI355X (gfx950) implementation of
 * FEABAS's two hot paths: the FFT cross-correlation matcher (feabas/matcher.py) and
 * the finite-element mesh relaxation (feabas/optimizer.py + mesh.py + material.py).
 *
 * The reference has no FFI of its own (it is pure Python over numpy/scipy); the
 * boundary it offers is the Python function surface of feabas.matcher /
 * feabas.optimizer / feabas.mesh.  Each entry point below names the reference
 * function (file:line under feabas/) whose arithmetic it replaces; the Python
 * package feabas_amd binds them with ctypes and keeps the reference signatures
 * (INTEGRATION.md shows the stub a FEABAS maintainer would add).
 *
 * Conventions
 *  - handle based: one fb_ctx per process per GPU, no global state;
 *  - every call returns 0 on success or a negative fb_status; the message is
 *    retrievable with fb_last_error(ctx); nothing throws across the boundary;
 *  - "host" entry points take caller-owned host buffers and are synchronous;
 *    "_dev" entry points take device pointers, enqueue on the context stream and
 *    return without synchronising (call fb_sync);
 *  - images are C-contiguous, row-major; DoF order is [x0,y0,x1,y1,...]
 *    (material.py:155, optimizer.py:124).
 */
#ifndef FEABAS_HIP_H
#define FEABAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fb_ctx fb_ctx;

typedef enum fb_status_e {
    FB_OK = 0,
    FB_ERR_ARG = -1,        /* invalid argument / shape */
    FB_ERR_HIP = -2,        /* HIP runtime error */
    FB_ERR_FFT = -3,        /* rocFFT error */
    FB_ERR_NOMEM = -4,
    FB_ERR_NOCONV = -5,     /* solver did not reach the requested residual */
    FB_ERR_BREAKDOWN = -6,  /* PCG breakdown: p^T A p <= 0 (matrix not PSD) */
    FB_ERR_COMM = -7
} fb_status;

/* constant.py:39-41 */
#define FB_CONF_NONE 0
#define FB_CONF_STD 1
#define FB_CONF_MIRROR 2

/* image element types accepted by fb_dog */
#define FB_U8 0
#define FB_F32 1

/* ------------------------------------------------------------------ context */
fb_ctx* fb_create(int device_id);
void fb_destroy(fb_ctx* ctx);
const char* fb_last_error(fb_ctx* ctx);
int fb_sync(fb_ctx* ctx);
void* fb_stream(fb_ctx* ctx);                 /* hipStream_t of the context */
int fb_device_info(fb_ctx* ctx, char* name, int name_len, int* num_cu, size_t* hbm_bytes);
const char* fb_version(void);

/* device memory owned by the context (freed by fb_free or fb_destroy) */
int fb_malloc(fb_ctx* ctx, size_t bytes, void** dptr);
int fb_free(fb_ctx* ctx, void* dptr);
/* page-locked host staging memory for fb_memcpy_h2d / fb_memcpy_d2h at the link rate */
int fb_host_alloc(fb_ctx* ctx, size_t bytes, void** hptr);
int fb_host_free(fb_ctx* ctx, void* hptr);
/* gather n row-major uint8 images (hs[k] x ws[k], row pitch pitches[k] bytes) into the slots of a host stack [n][H][W]
 * (top-left corners) with `threads` host threads: the packing of cropped strips into the staging buffer */
int fb_host_pack2d(fb_ctx* ctx, uint8_t* dst, int n, int H, int W, const void* const* srcs, const int* hs, const int* ws,
                   const int64_t* pitches, int threads);
int fb_memcpy_h2d(fb_ctx* ctx, void* dst, const void* src, size_t bytes);
int fb_memcpy_d2h(fb_ctx* ctx, void* dst, const void* src, size_t bytes);
/* device -> device on the context's stream (asynchronous, ordered with the kernels) */
int fb_memcpy_d2d(fb_ctx* ctx, void* dst, const void* src, size_t bytes);
/* `rows` rows of `width_bytes` bytes between two pitched device arrays (an image inside its slot <-> a dense copy of it), on the
 * context's stream */
int fb_memcpy2d_d2d(fb_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows);
int fb_memset(fb_ctx* ctx, void* dst, int value, size_t bytes);

/* HIP-event stopwatch on the context stream (bench.py's timed region) and
 * per-kernel accumulators.  fb_prof_enable(1) brackets every launch of the
 * library's own kernels with events; fb_prof_get returns the number of launches,
 * their summed duration and the algorithmic HBM bytes they moved. */
int fb_timer_start(fb_ctx* ctx);
int fb_timer_stop(fb_ctx* ctx, float* ms);
int fb_prof_enable(fb_ctx* ctx, int on);
int fb_prof_reset(fb_ctx* ctx);
int fb_prof_count(fb_ctx* ctx);
int fb_prof_get(fb_ctx* ctx, int index, char* name, int name_len, int* launches, double* total_ms,
                double* total_algorithmic_bytes);

/* ------------------------------------------------------------------ NCC path */
/* next 5-smooth length, scipy.fftpack.next_fast_len as used at matcher.py:59-62 */
int fb_next_fast_len(int n);

/* matcher.xcorr_fft (matcher.py:22-135) with sigma=0, normalize=False:
 * img0 [N][C][H0][W0], img1 [N][C][H1][W1] float32 (C = 1 for 3-D stacks; the
 * reference's (N,H,W,C) input must be moved to channel-major by the caller, as
 * matcher.py:50-53 does).  Outputs dx, dy (float64[N]) and conf (float32[N]):
 * centre(img1) + (dx,dy) <-> centre(img0).  Integer peaks follow numpy's
 * first-maximum rule (matcher.py:82). */
int fb_ncc_batch(fb_ctx* ctx, const float* img0, const float* img1, int N, int C,
                 int H0, int W0, int H1, int W1, int pad, int subpixel, int conf_mode,
                 double* dx, double* dy, float* conf);
int fb_ncc_batch_dev(fb_ctx* ctx, const float* img0, const float* img1, int N, int C,
                     int H0, int W0, int H1, int W1, int pad, int subpixel, int conf_mode,
                     double* dx, double* dy, float* conf);
/* matcher.xcorr_fft(normalize=True) (matcher.py:70-81, 119-122; no reference call site enables it): as fb_ncc_batch, with the
 * correlation surface divided by NC = irfft2(conj(M0) M1) of the two masks -- scaled by its maximum (at least 1) and clipped at
 * 0.1 -- before the peak, the sub-pixel fit and the confidence look at it, and the mirror surface by the same of irfft2(M0 M1).
 * mask0 [H0][W0], mask1 [H1][W1]: host float32, one pair for the whole batch; NULL = all ones.  Runs on the rocFFT class (the
 * surfaces exist in memory there) whatever the shape. */
int fb_ncc_batch_normalized(fb_ctx* ctx, const float* img0, const float* img1, int N, int C,
                            int H0, int W0, int H1, int W1, const float* mask0, const float* mask1,
                            int pad, int subpixel, int conf_mode, double* dx, double* dy, float* conf);
/* Block matching straight from resident image stacks: the translation-only form of
 * MeshRenderer.crop_multiple + xcorr_fft (matcher.py:834-846).  imgs0 [P][IH0][IW0], imgs1
 * [P][IH1][IW1] float32 (DoG output); blk [N][9] int32 = {image, x0, y0, h0, w0, x1, y1, h1, w1}:
 * block n correlates the h0 x w0 window at (x0,y0) of imgs0[image] with the h1 x w1 window at
 * (x1,y1) of imgs1[image]; pixels outside an image read 0 (dal.StreamLoader fillval).  All blocks
 * of a call share the FFT shape (Fh, Fw) the caller derives with the rule of matcher.py:59-62;
 * hmax, wmax = upper bounds of the block heights / widths (0 = unknown). */
int fb_ncc_blocks_dev(fb_ctx* ctx, const float* imgs0, const float* imgs1, int IH0, int IW0, int IH1, int IW1,
                      int N, const int* blk, int hmax, int wmax, int Fh, int Fw, int subpixel, int conf_mode,
                      double* dx, double* dy, float* conf);
/* As fb_ncc_blocks_dev, but the window of image 1 is gathered through a per-block affine map with the bilinear rule of
 * cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0): the affine-approximated branch of MeshRenderer.crop_multiple
 * (renderer.py:419-451, 499-511 -> common.render_by_subregions, common.py:218-350) that
 * matcher.bboxes_mesh_renderer_matcher (matcher.py:833-846) takes for a deformed mesh1.
 * aff1: device double [N][10] = {x0, y0, A00, A10, t0, A01, A11, t1, xmin, ymin}: output pixel (i, j) of block n reads
 * image 1 at (X A00 + Y A10 + t0, X A01 + Y A11 + t1), X = x0 + i, Y = y0 + j; (xmin, ymin) = integer origin of the
 * sub-image the reference passes to cv2.remap (the float32 map is relative to it).  blk[n] = {image, x0, y0, h0, w0,
 * (unused) x1, y1, h1, w1}.  NULL aff1 = fb_ncc_blocks_dev. */
int fb_ncc_blocks_affine_dev(fb_ctx* ctx, const float* imgs0, const float* imgs1, int IH0, int IW0, int IH1, int IW1, int N,
                             const int* blk, const double* aff1, int hmax, int wmax, int Fh, int Fw, int subpixel, int conf_mode,
                             double* dx, double* dy, float* conf);
/* debugging / parity aid: the two correlation surfaces (un-normalised) of the
 * last fb_ncc_batch* call that went through the streaming (rocFFT) class. */
int fb_ncc_last_surfaces(fb_ctx* ctx, float* C_out, float* Cm_out, int* Fh, int* Fw);

/* common.masked_dog_filter (common.py:353-377).  img: [N][H][W] uint8 or float32;
 * mask: [H][W] uint8 (0 = outside) or NULL; out: [N][H][W] float32. */
int fb_dog(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, double sigma,
           const uint8_t* mask, int signed_out, float* out);
int fb_dog_dev(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, double sigma,
               const uint8_t* mask, int signed_out, float* out);

/* cv2.resize(fx=fy=0.5, INTER_AREA) of even-sized uint8 images (matcher.py:255-256) */
int fb_area_downsample2(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, uint8_t* out);
int fb_area_downsample2_dev(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, uint8_t* out);
/* cv2.resize(img, None, fx, fy, INTER_AREA) of uint8 images for any shrinking factor 0 < f <= 1 (the coarse_downsample /
 * fine_downsample options of stitching_matcher, matcher.py:254-266, 318-335): output fb_area_resize_size(H, fy) x
 * fb_area_resize_size(W, fx) = cvRound(n f); integer 1 / f: the integer cell sums of resizeAreaFast, otherwise the
 * fractional-coverage taps of computeResizeAreaTab accumulated in float.  fb_area_resize: host arrays. */
int fb_area_resize_size(int n, double f);
int fb_area_resize_dev(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, double fx, double fy, uint8_t* out);
int fb_area_resize(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, double fx, double fy, uint8_t* out);
/* Images of unequal size in one padded stack (the strips of a real section differ in shape from pair to pair,
 * stitcher.py:561-571): image n is the sizes[n] = {h, w} corner (device int32 [N][2]) of its H x W slot and is
 * downsampled / filtered as an h x w image; the downsampled stack has half_size(H) x half_size(W) slots. */
int fb_area_downsample2_sizes_dev(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, const int* sizes, uint8_t* out);
int fb_dog_sizes_dev(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, const int* sizes, double sigma, int signed_out,
                     float* out);
/* common.masked_dog_filter with one mask per image (masks uint8 [N][H][W], device): the DoG MeshRenderer.crop_multiple
 * applies to its N x h x w stack when log_sigma > 0 (renderer.py:632-641); np.ptp over the whole stack, as there. */
int fb_dog_masks_dev(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, double sigma, const uint8_t* masks,
                     int signed_out, float* out);

/* common.masked_dog_filter(cv2.resize(img, fx=fy=0.5, INTER_AREA), sigma) (matcher.py:255-256 + 273-274) of N resident uint8
 * images [N][H2][W2] in one kernel: the 2 x 2 cells are averaged in the DoG's loader, the coarse image is never written.
 * out float32 [N][half_size(H2)][half_size(W2)], half_size = cvRound(n / 2). */
int fb_dog_down2_dev(fb_ctx* ctx, const uint8_t* img, int N, int H2, int W2, double sigma, int signed_out, float* out);
/* the two strip stacks of a batch of pairs (matcher.py:273-274, 336-337: both images of a pair get the same filter) in ONE
 * launch: images 0 .. N-1 from img0, N .. 2N-1 from img1, out [2N] */
int fb_dog_pair_dev(fb_ctx* ctx, const void* img0, const void* img1, int dtype, int N, int H, int W, double sigma, int signed_out, float* out);
int fb_dog_down2_pair_dev(fb_ctx* ctx, const uint8_t* img0, const uint8_t* img1, int N, int H2, int W2, double sigma, int signed_out, float* out);

/* mask[i] &= (lo <= img[i] <= hi) on device arrays: the `mask_range` of MeshRenderer.crop_multiple (renderer.py:634-637),
 * applied to the rendered stack before its masked DoG */
int fb_mask_range_dev(fb_ctx* ctx, const float* img, size_t n, float lo, float hi, uint8_t* mask);
/* number of non-zero bytes of a device array, returned to the host: the `mask.any()` test of MeshRenderer.crop_multiple
 * (renderer.py:601-648 returns None for a stack no pixel of which is covered; matcher.py:835-839 then skips the batch)
 * without bringing the N x h x w mask across the link */
int fb_count_nonzero_dev(fb_ctx* ctx, const uint8_t* m, size_t n, int64_t* count);

/* MeshRenderer for general triangulated meshes (one region, no collisions), all pointers device pointers.
 * fb_mesh_candidates_dev: for NB blocks of h x w pixels whose first pixel sits at org [NB][2] (float64, MOVING coordinates
 *   with the mesh offset removed, renderer.py:286-289), the triangles (tris int32 [T][3] over v_mov float64 [V][2]) whose
 *   box meets the closed box bbox - 0.5 (the STRtree query of renderer.py:405): cand [NB][cap], count [NB] (count may
 *   exceed cap: the caller retries with a larger cap).
 * fb_mesh_render_blocks_dev: MeshRenderer.crop_multiple(bboxes, mode=RENDER_FULL, log_sigma=0, remap_interp=INTER_LINEAR)
 *   (renderer.py:601-631): tier [NB] 1 / 2 = affine field A6 [NB][6] = {A00, A10, t0, A01, A11, t1} (crop_field_affine,
 *   renderer.py:419-451), 3 = exact piecewise-linear field through the candidates (field_w_weight, renderer.py:259-300);
 *   v_img [V][2] = image-space vertices (INITIAL gear with offset).  The field is sampled from the resident image
 *   (dtype FB_U8: cv2's fixed-point bilinear, FB_F32: its float path; pixel (0, 0) of the array at (img_x0, img_y0) in
 *   image space, 0 outside = StreamLoader fillval + BORDER_CONSTANT) through float32 maps relative to origin [NB][2]
 *   (out; = floor(min field) - 4 of the whole stack while it spans < 16300 px, common.py:264, 305-321, else per block).
 *   ext int32 [NB][4] = scratch (field extent per block).  out float32 [NB][h][w], mask uint8 [NB][h][w]. */
/* fb_mesh_block_affines (host arrays, no device work): bbox_affine_tform + the tolerance test of crop_field
 *   (renderer.py:397-416, 499-511) for every block from its candidate list: tier [NB] 2 = block affine (A6), 3 = exact field,
 *   -1 = degenerate / flipped fit (caller's statement-by-statement route). */
int fb_mesh_block_affines(fb_ctx* ctx, int V, const double* v_mov, const double* v_img, const int32_t* tris, int NB, const double* org,
                          int h, int w, int cap, const int32_t* cand, const int32_t* count, double tol, int32_t* tier, double* A6);
/* fb_mesh_block_uncovered (host arrays, no device work): the area of every block's box that no candidate triangle covers (MOVING
 *   coordinates) -- the quantity the precise mask of crop_field_affine compares with 1 px^2 (renderer.py:437-447).  Blocks of
 *   tier 1 / 2 with uncovered >= 1 are rendered with tier 11 / 12: the affine field, masked where the pixel lies outside the
 *   mesh (fb_mesh_render_blocks_dev). */
int fb_mesh_block_uncovered(fb_ctx* ctx, int V, const double* v_mov, const int32_t* tris, int NB, const double* org, int h, int w, int cap,
                            const int32_t* cand, const int32_t* count, double* uncovered);
/* fb_mesh_block_uncovered_dev: the same quantity from the device-resident vertex / triangle / candidate arrays of
 *   fb_mesh_candidates_dev (uncovered: device float64 [NB]). */
int fb_mesh_block_uncovered_dev(fb_ctx* ctx, const double* v_mov, const int* tris, int NB, const double* org, int h, int w, int cap, const int* cand,
                                const int* count, double* uncovered);
/* fb_signed_area (host arrays, no device work): common.signed_area (common.py:672-676) -- twice the signed area
 *   cross(p1 - p0, p2 - p1) of the T triangles tris [T][3] over v [V][2] (negative indices count from the end, as numpy
 *   takes them), rounded operation by operation like the numpy statement. */
int fb_signed_area(fb_ctx* ctx, int V, const double* v, int T, const int32_t* tris, double* area);
/* fb_tri_edge_ratio (host arrays, no device work): the gathers of Mesh.triangle_edge_deform (mesh.py:1966-1976):
 *   ratio [T][3] = squared length of every triangle edge in v1 over its squared length in v0 (edge k: vertex k - 1 -> vertex k). */
int fb_tri_edge_ratio(fb_ctx* ctx, int V, const double* v0, const double* v1, int T, const int32_t* tris, double* ratio);
/* fb_mesh_locate_dev: Mesh.tri_finder (mesh.py:2080-2188) for K points pts [K][2] (device, frame of v_mov): tid [K] = the
 *   containing triangle of smallest index, -1 outside the mesh. */
int fb_mesh_locate_dev(fb_ctx* ctx, int T, const double* v_mov, const int* tris, int K, const double* pts, int* tid);
int fb_mesh_candidates_dev(fb_ctx* ctx, int T, const double* v_mov, const int* tris, int NB, const double* org, int h, int w, int cap,
                           int* cand, int* count);
int fb_mesh_render_blocks_dev(fb_ctx* ctx, const void* img, int dtype, int IH, int IW, int img_x0, int img_y0, const double* v_mov,
                              const double* v_img, const int* tris, int NB, const double* org, int h, int w, const int* tier,
                              const double* A6, int cap, const int* cand, const int* count, int* ext, int* origin, float* out,
                              uint8_t* mask);

/* common.remap = cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) (common.py:218-255, called at 329-330) of resident float32
 * images through explicit per-pixel maps: the exact piecewise-linear tier of MeshRenderer.crop_multiple
 * (renderer.py:511-563).  All pointers are device pointers.  imgs [P][IH][IW]; img_id [N]; map_x, map_y float32
 * [N][h][w] relative to origin [N][2] = the integer (xmin, ymin) of the sub-image of render_by_subregions
 * (common.py:316-321); mask uint8 [N][h][w] (nullable; 0 = fill value 0); out float32 [N][h][w]. */
int fb_remap_dev(fb_ctx* ctx, const float* imgs, int IH, int IW, int N, const int* img_id, int h, int w, const float* map_x,
                 const float* map_y, const uint8_t* mask, const int* origin, float* out);

/* Calibration of the HBM roofline (bench support, not a reference function): GB/s of a plain streaming kernel reading nr
 * and writing nw unit-stride streams of 1 GiB each (nr, nw in 0..2).  The 8 TB/s of the data sheet is a read figure; the NCC
 * passes write as much as or twice what they read. */
int fb_hbm_probe(fb_ctx* ctx, int nr, int nw, double* gbs);
/* Synthetic overlap strips for benchmarks (not a reference function): pair p gets an integer
 * offset (sx, sy), multiples of shift_step in [-max_shift, max_shift]^2, derived from (seed, pair0 + p); strips0/strips1 are
 * uint8 [P][H][W] with strips1(x, y) = texture(x + sx + wx, y + sy + wy), (wx, wy) a smooth warp of amplitude `warp` px
 * (<= 0.4); shifts_dev receives [P][2] = {sx, sy}. */
int fb_synth_strips_dev(fb_ctx* ctx, int P, int pair0, int H, int W, uint32_t seed, int max_shift, int shift_step, float warp,
                        uint8_t* strips0, uint8_t* strips1, int* shifts_dev);

/* ------------------------------------------------------------------ FEM path */
/* A spring-linked-mesh system (optimizer.SLM, optimizer.py:487-1873) resident on the
 * GPU.  Free (unlocked) meshes occupy consecutive ranges of a global free-vertex
 * numbering, DoF = 2*vertex + {0,1} exactly like SLM.index_offsets
 * (optimizer.py:960-970); locked meshes only appear through the links' residuals.
 *
 *   fb_sys_create -> fb_sys_add_mesh (per free mesh) -> fb_sys_set_links -> fb_sys_finalize
 *   (symbolic: one 2x2 block per coupled vertex pair)
 *   fb_sys_assemble_mesh  : Mesh.stiffness_matrix (mesh.py:3058-3083) with the linear
 *                           engineering element of material.py:134-182, times soft_factor
 *                           (optimizer.py:821-822); stress = K (v_cur - v_shape) as float32
 *   fb_sys_assemble_links : SLM.crosslink_terms (optimizer.py:832-901)
 *   fb_sys_lambda         : SLM.relative_lambda_trace (optimizer.py:1573-1590)
 *   fb_sys_form           : A = ls K + lc C, b = lc rhs - ls stress (optimizer.py:1417-1418)
 *   fb_sys_solve          : optimizer.solve (optimizer.py:1945-2080)
 */
typedef struct fb_system fb_system;
int fb_sys_create(fb_ctx* ctx, int64_t nvert_free, fb_system** out);
void fb_sys_destroy(fb_ctx* ctx, fb_system* sys);
/* tri: [T][3] vertex ids local to the mesh; the mesh owns global vertices [voff, voff+V) */
int fb_sys_add_mesh(fb_ctx* ctx, fb_system* sys, int64_t voff, const int32_t* tri, int V, int T, int* mesh_id);
/* nodes6: [K][6] global free-vertex ids of the two triangles a match sits in
 * (slots 0-2 side 0, slots 3-5 side 1); -1 marks a vertex of a locked mesh */
int fb_sys_set_links(fb_ctx* ctx, fb_system* sys, int64_t K, const int32_t* nodes6);
int fb_sys_finalize(fb_ctx* ctx, fb_system* sys, int64_t* nnzb);
int fb_sys_pattern(fb_ctx* ctx, fb_system* sys, int64_t* browptr /*[nv+1]*/, int32_t* bcol /*[nnzb]*/);
int fb_sys_info(fb_ctx* ctx, fb_system* sys, int64_t* nv, int64_t* nnzb, int64_t* nlink);
/* v_shape, v_cur: [V][2] float64 (v_cur NULL = zero stress); tri_mult: [T] float32 or NULL
 * (mesh.stiffness_multiplier * material.stiffness_multiplier) */
int fb_sys_assemble_mesh(fb_ctx* ctx, fb_system* sys, int mesh_id, const double* v_shape, const double* v_cur,
                         const float* tri_mult, double nu, double soft);
/* Same, but ADDS to the rows: for meshes entered at the same vertex offset as an earlier mesh (the `groupings` of
 * SLM.optimize_linear, optimizer.py:1378-1415, make the members of a group share their degrees of freedom). */
int fb_sys_assemble_mesh_add(fb_ctx* ctx, fb_system* sys, int mesh_id, const double* v_shape, const double* v_cur,
                             const float* tri_mult, double nu, double soft);
/* Mixed materials (Mesh.stiffness_matrix with non-engineering elements, mesh.py:2992-3083;
 * element maths material.py:185-309): per-triangle model (0 engineering-linear, 1 St-Venant-Kirchhoff,
 * 2 Neo-Hookean), Poisson ratio and material stiffness multiplier.  K is the tangent stiffness at v_cur,
 * stress = K_lin (v_cur - v_shape) + internal force of the non-linear elements, float32. */
int fb_sys_assemble_mesh_materials(fb_ctx* ctx, fb_system* sys, int mesh_id, const double* v_shape, const double* v_cur,
                                   const float* tri_mult, const int32_t* tri_model, const double* tri_nu,
                                   const float* tri_matmult, double soft);
/* Materials whose stiffness follows the area stretch of the triangle (Material._stiffness_func: Mesh.
 * nonlinear_engineering_stiffness_matrix, mesh.py:2937-2971, for engineering elements; the f(J) modifier of material.py:307-308 for the
 * others; material.asymmetrical_elasticity, material.py:546-551 -- the default "wrinkle" material, default_material_table.yaml:46-56).
 * As fb_sys_assemble_mesh_materials, plus: v_init [V][2] the INITIAL gear; tri_func [T] the stiffness function of the triangle's
 * material (-1 = none); function k is piecewise linear through the knots (func_x, func_y)[func_ptr[k] .. func_ptr[k+1]) (x ascending,
 * >= 2 knots), constant beyond its ends, and is evaluated at (area(v_cur) / area(v_init)) / base, base = sum|area(v_cur)| /
 * sum|area(v_init)| over the linear triangles (engineering, no function; all triangles if there is none), mesh.py:2952-2963.
 * func_matmult [nfunc]: the material multiplier of function k's material in double precision (the reference multiplies it in
 * double for the non-engineering elements; engineering elements take the float32 tri_matmult as without a function). */
int fb_sys_assemble_mesh_stretch(fb_ctx* ctx, fb_system* sys, int mesh_id, const double* v_shape, const double* v_cur, const double* v_init,
                                 const float* tri_mult, const int32_t* tri_model, const double* tri_nu, const float* tri_matmult,
                                 const int32_t* tri_func, int nfunc, const int32_t* func_ptr, const double* func_x, const double* func_y,
                                 const double* func_matmult, double soft);
/* bary6: [K][6] = [+B0 | -B1] (Link.shape_matrix_contrib, optimizer.py:114-131); w: [K] float32
 * (weight * residue_weight); rxy: [K][2] residual x1 - x0 (Link.dxy, optimizer.py:248-255) */
int fb_sys_assemble_links(fb_ctx* ctx, fb_system* sys, const double* bary6, const float* w, const double* rxy);
/* Replace the links of a finalized system without touching the symbolic pattern (the new matches must
 * couple only vertices that are already coupled, e.g. matches against locked meshes). */
int fb_sys_update_links(fb_ctx* ctx, fb_system* sys, int64_t K, const int32_t* nodes6);
/* Link.xy0 / xy1 / dxy (optimizer.py:121-135, 248-255) for the K matches of one link, laid out as fb_sys_set_links /
 * fb_sys_assemble_links take them (host only): tri [T][3], v [V][2] of the two meshes at the gears of the solve, tid [K] int64,
 * B [K][3]; voff = first global free vertex of the mesh, < 0 when it is locked; (ox, oy) = offset(mesh1) - offset(mesh0).
 * nodes6 [K][6]; bary6 [K][6] = [B0 | -B1] and rxy [K][2] = xy1 - xy0 may be NULL. */
int fb_link_terms(fb_ctx* ctx, int64_t K, const int32_t* tri0, int64_t T0, const double* v0, const int64_t* tid0, const double* B0, int64_t voff0,
                  const int32_t* tri1, int64_t T1, const double* v1, const int64_t* tid1, const double* B1, int64_t voff1, double ox, double oy,
                  int32_t* nodes6, double* bary6, double* rxy);
/* Block-diagonal batch of `ngroups` independent systems stored as equal consecutive vertex ranges (one
 * tile pair each, matcher.py:551): relative_lambda_trace per range, then A and b as in fb_sys_form. */
int fb_sys_form_groups(fb_ctx* ctx, fb_system* sys, int ngroups, double stiffness_lambda, double crosslink_lambda,
                       double* ls_out);
/* Solve every range of a block-diagonal batch (after fb_sys_form_groups) with its own Jacobi-PCG, one workgroup
 * per range, to ||A x - b|| <= max(rtol, atol/||b||) ||b|| of THAT range -- the in-matcher relaxations of
 * matcher.py:717-742 for a whole batch of tile pairs in one launch.  x: host [2 nv]; iters_max / relres_max: worst range. */
int fb_sys_solve_groups(fb_ctx* ctx, fb_system* sys, int ngroups, double* x, double rtol, double atol, int maxiter, int precond,
                        int* iters_max, double* relres_max);
/* x^T K x per equal vertex range (K of the last fb_sys_assemble_mesh*): Es and Es0 of the strain estimate,
 * matcher.py:764-777.  x: host [2 nv], energy: host [ngroups]. */
int fb_sys_group_energy(fb_ctx* ctx, fb_system* sys, int ngroups, const double* x, double* energy);
/* ---- batched tile-pair stages (the FEM work inside matcher.iterative_xcorr_matcher_w_mesh for P pairs at once).
 * sys: P copies of the cartesian mesh of matcher.py:354-359 (nx x ny nodes at xs, ys; cells (a b / c d) split into
 * (a, b, d), (a, d, c)) entered as one mesh, no links at finalize; mesh0 of every pair is locked.
 * fb_pairs_relax  : matcher.py:725-737 -- optimize_linear of every pair against its K matches, then the huber residue
 *                   weight rw[K] of every match (optimizer.py:174-205).  x_out (nullable): vertex field [2 nv].
 * fb_pairs_strain : matcher.py:752-777 after the rigid fit R [P][3][3] of every pair (spatial.fit_affine, host). */
int fb_pairs_relax(fb_ctx* ctx, fb_system* sys, int P, int nx, int ny, const double* xs, const double* ys, int64_t K, const int32_t* pid,
                   const double* xy0_moving, const double* xy1_initial, const double* t1, const float* conf, double residue_len,
                   int residue_mode /* 0 huber, 1 threshold (optimizer.py:198-205) */, double sample_err, double stiffness_lambda,
                   double rtol, float* rw, double* x_out, int* iters, double* relres);
/* fb_pairs_relax for matches located in a DEFORMED mesh1 (a pair whose earlier relaxation was not a rigid translation;
 * Link.from_coordinates on the MOVING gear, optimizer.py:51-82 -> mesh.py:2191-2217): nodes3 [K][3] = the mesh1 vertices
 * of every match (ids inside the union mesh), B1 [K][3] its barycentric coordinates, dxy0 [K][2] = the link residual
 * with mesh1 at its FIXED gear.  x_out = TOTAL displacement from the FIXED gear (the stress term of
 * optimizer.py:1417-1418 is inside the system). */
int fb_pairs_relax_bary(fb_ctx* ctx, fb_system* sys, int P, int64_t K, const int32_t* nodes3, const double* B1, const double* dxy0,
                        const float* conf, double residue_len, int residue_mode, double sample_err,
                        const double* sample_err_each /* [K] or NULL */, double stiffness_lambda, double rtol, float* rw, double* x_out,
                        int* iters, double* relres);
/* fb_pairs_strain for pairs whose meshes share a topology but not a geometry (strips of unequal size): matches located by
 * the caller (nodes3, B1 in the INITIAL mesh of their pair), es0 [P] = v0^T K v0 of every pair's centred mesh. */
int fb_pairs_strain_bary(fb_ctx* ctx, fb_system* sys, int P, int64_t K, const int32_t* pid, const int32_t* nodes3, const double* B1,
                         const double* xy0_fixed, const double* xy1_initial, const float* weight, const double* R, double stiffness_lambda,
                         const double* es0, double default_strain, double* strain, int* iters, double* relres);
/* ---- matcher.stitching_matcher (matcher.py:224-367) for a batch of P equal-shaped overlap-strip pairs resident in HBM,
 * as one entry: what stitcher.py:593 calls once per overlap (downsample, DoG, global NCC, fine DoG, coarse-to-fine block
 * NCC rounds with the pad / subpixel / spacing schedule, rigid relaxations between rounds, last-round relaxation + residue
 * weights, strain).  strips0 / strips1: device uint8 [P][H][W].  Per pair: tx, ty (global translation at full resolution),
 * conf0, valid, flags, strain (DEFAULT_AVG_DEFORM where there is none).  The match table (rows of a pair contiguous;
 * xy0 / xy1 in the INITIAL gears of mesh0 / mesh1 like stitching_matcher's return value) stays in the matcher until
 * fb_match_strips_table copies its *nrows rows out.  Pairs with flags != 0 are NOT finished here (valid = 0, no rows):
 * they need a branch of the reference that works pair by pair (second shot of global_translation_matcher, a deformed
 * mesh1 between spacings, relax_first, the degenerate branches of fit_affine) -- the caller's general route
 * (feabas_amd.stitch_pipeline) takes them.  Masks and the photometric statistics are on that route too. */
typedef struct fb_strip_matcher fb_strip_matcher;
typedef struct fb_strip_opts {
    double sigma;              /* DoG sigma at full resolution (matcher.py:233) */
    int coarse_downsample2;    /* 1: coarse_downsample = 0.5, 0: coarse_downsample = 1 (matcher.py:234) */
    double conf_thresh;        /* matcher.py:232 */
    int min_num_blocks;        /* last round (matcher.py:572) */
    int conf_mode;             /* FB_CONF_* */
    double residue_len;        /* pixels at full resolution; <= 0: no residue filter (matcher.py:236) */
    int residue_mode;          /* 0 huber, 1 threshold (matcher.py:730-735) */
    double stiffness_lambda;   /* matcher.py:507 */
    double relax_tol;          /* PCG tolerance of the last-round relaxation */
    int compute_strain;        /* matcher.py:497 */
    int nspacings;             /* 0: the automatic spacings of matcher.py:243-251 */
    const double* spacings;    /* pixels (>= 1) */
} fb_strip_opts;
#define FB_STRIP_LOWCONF 1      /* not reported any more: the second shot of global_translation_matcher (matcher.py:159-221) runs inside the entry */
#define FB_STRIP_NONRIGID 2     /* not reported any more: the deformed-mesh branch (matcher.py:725-742, 833-846) runs inside the entry */
#define FB_STRIP_RELAXFIRST 4
#define FB_STRIP_RIGIDFIT 8
#define FB_STRIP_FOLDED 16      /* a block of a deformed mesh1 has a degenerate / flipped affine fit (renderer.py:397-416) */
/* common.divide_bbox (feabas/common.py:380-409), host only: the bounding box bbox = {xmin, ymin, xmax, ymax} cut into
 * max(ceil(extent / block), min_blocks) blocks per axis (block_hw = {h, w}, min_blocks_yx = {ny, nx}) of ceil(extent / count)
 * pixels, starts = numpy.linspace(lo, hi - step, count), a shrink_factor != 1 scales the blocks about their centres, round_output
 * rounds the starts half to even (np.round).  counts_xy = {nx, ny}, steps_xy = {step_x, step_y}; x_start / y_start (capacity cap_x /
 * cap_y doubles, may be NULL for a sizing call).  The block grid fb_match_strips walks is made by the same code. */
int fb_divide_bbox(fb_ctx* ctx, const double* bbox, const double* block_hw, const int* min_blocks_yx, double shrink_factor, int round_output,
                   int* counts_xy, int* steps_xy, double* x_start, int cap_x, double* y_start, int cap_y);
/* Round stepper of the coarse-to-fine block matcher (iterative_xcorr_matcher_w_mesh, matcher.py:567-716), host only: which
 * spacing the next round of block matching runs at and whether its blocks are zero padded, given the largest displacement the
 * last round measured.  create: the spacings (any order), allow_enlarge / allow_dwell / max_spacing_skip as the reference's
 * keywords, pad_fixed -1 = by rule (1 / 0: the `pad` keyword).  round: 0 when the walk is over, else 1 and the spacing, whether it
 * is the last (smallest) one and the padding.  advance: after a round with largest displacement max_dis (multiplier = the
 * reference's min_block_size_multiplier, 4); *redo = 1 when that round has to be repeated at the enlarged spacing before any link
 * is made (matcher.py:693-699). */
typedef struct fb_schedule fb_schedule;
fb_schedule* fb_schedule_create(const double* spacings, int n, int allow_enlarge, int allow_dwell, int max_spacing_skip, int pad_fixed);
void fb_schedule_destroy(fb_schedule* s);
int fb_schedule_round(const fb_schedule* s, double* spacing, int* last, int* pad);
int fb_schedule_advance(fb_schedule* s, double max_dis, double multiplier, int* redo);
int fb_strip_matcher_create(fb_ctx* ctx, int P, int H, int W, const fb_strip_opts* opts, fb_strip_matcher** out);
/* Strips of unequal size (the usual case in a real section: every overlap follows the stage jitter of its two tiles,
 * stitcher.py:561-571): pair p is the shapes[p] = {h, w} top-left corner of its H x W slot; every stage works on the pair's own
 * extent, with its own block grids, spacings (automatic ones per shape) and mesh geometry.  The pairs of a batch must share the
 * number of spacings and the node grid of Mesh.from_bbox (feabas_amd.stitch_pipeline.RaggedStripBatchMatcher.bucket_key). */
int fb_strip_matcher_create_ragged(fb_ctx* ctx, int P, int H, int W, const int32_t* shapes, const fb_strip_opts* opts, fb_strip_matcher** out);
void fb_strip_matcher_destroy(fb_ctx* ctx, fb_strip_matcher* m);
int fb_strip_matcher_info(fb_ctx* ctx, fb_strip_matcher* m, int* nspacings, double* spacings, int* grid_nx, int* grid_ny,
                          int* relax_iters, double* relax_relres, int* strain_iters, double* strain_relres);
int fb_match_strips(fb_ctx* ctx, fb_strip_matcher* m, const uint8_t* strips0, const uint8_t* strips1, double* tx, double* ty,
                    float* conf0, uint8_t* valid, uint8_t* flags, double* strain, int64_t* nrows);
int fb_match_strips_table(fb_ctx* ctx, fb_strip_matcher* m, int32_t* pair, double* xy0, double* xy1, float* weight);
/* Extras of the NEXT fb_match_strips call on this matcher: masks0 / masks1 = P host pointers each (the
 * array or an entry may be NULL) to uint8 valid-pixel masks, non-zero = valid -- [H][W] on a matcher of equal strips, one
 * contiguous array of the pair's own shape [Hs[p]][Ws[p]] on a ragged one (round 6) -- the masked DoG of both scales
 * (matcher.py:257-274, 336-337; common.py:353-377); photometric != 0 (equal strips only): the statistics of matcher.py:279-314, read afterwards
 * with fb_match_strips_photometric: phtm [P][4] = mean grey level of the two coarse strips and mean |DoG| of the two filtered
 * ones over the overlap of the translated strips, has [P] = 0 where strip 0 has fewer than 4 valid pixels there (None). */
int fb_strip_matcher_set_extras(fb_ctx* ctx, fb_strip_matcher* m, const uint8_t* const* masks0, const uint8_t* const* masks1, int photometric);
int fb_match_strips_photometric(fb_ctx* ctx, fb_strip_matcher* m, double* phtm, uint8_t* has);
/* The deformed-mesh branch of the last fb_match_strips call (matcher.py:725-742): deformed [P] = the relaxation of mesh1
 * between two spacings was not a rigid translation; ntiers [P] = number of blocks of the pair's last deformed round; *nodes =
 * V, the nodes of one mesh (all nullable).  fb_match_strips_field: field [P][V][2] = MOVING - INITIAL of every mesh1 node
 * (zero for a translated grid), tiers = the MeshRenderer.crop_multiple tier of those blocks (1 global affine, 2 block affine,
 * 3 exact field), pairs concatenated in order (both nullable). */
int fb_match_strips_deformed(fb_ctx* ctx, fb_strip_matcher* m, uint8_t* deformed, int32_t* ntiers, int* nodes);
int fb_match_strips_field(fb_ctx* ctx, fb_strip_matcher* m, double* field, int32_t* tiers);
/* ---- host geometry of pairs whose mesh1 is deformed (no device work; ctx may be NULL).
 * fb_deformed_block_affines: the tier decision of MeshRenderer.crop_field with the affine approximator of
 *   MeshRenderer.from_mesh (renderer.py:90-109, 397-416, 453-511) for the nblk blocks of Q pairs.  vm [Q][nx ny][2] =
 *   MOVING vertices (with offset) of the cartesian mesh1 whose INITIAL nodes are xs x ys; bboxes [Q][nblk][4] int32;
 *   tol = affine_approx_tol.  tier [Q][nblk]: 1 global affine, 2 block affine, 3 exact field, -1 degenerate / flipped
 *   fit (caller's statement-by-statement route); A6 [Q][nblk][6] = {A00, A10, t0, A01, A11, t1}; lo [Q][2] = smallest
 *   image x / y an affine block samples (the remap origin of common.py:316-321 is floor(lo) - 4).
 * fb_deformed_locate: Mesh.tri_finder + cart2bary (mesh.py:2080-2217) on those meshes: point k of pair pair_of[k] ->
 *   tid [K] (cell (a b / c d): 2 cell = (a, b, d), 2 cell + 1 = (a, d, c); -1 outside) and B [K][3]. */
/* per_pair_grid != 0: xs [Q][nx], ys [Q][ny] -- every pair has its own node grid (strips of unequal size);
 * tol_each (nullable) [Q]: the tolerance of every pair (it follows the pair's spacing, matcher.py:599) */
int fb_deformed_block_affines(fb_ctx* ctx, int Q, int nx, int ny, const double* xs, const double* ys, int per_pair_grid, const double* vm,
                              int nblk, const int32_t* bboxes, double tol, const double* tol_each, int32_t* tier, double* A6, double* lo);
/* fb_deformed_exact_field: field_w_weight (renderer.py:259-300) for NB blocks of h x w pixels at org [NB][2] of pairs
 *   pair_of [NB]: the exact piecewise-linear inverse map (map_x, map_y float64 [NB][h][w]) and its mask (uint8). */
int fb_deformed_exact_field(fb_ctx* ctx, int Q, int nx, int ny, const double* xs, const double* ys, int per_pair_grid, const double* vm,
                            int NB, const int32_t* pair_of, const int32_t* org, int h, int w, double* map_x, double* map_y, uint8_t* mask);
int fb_deformed_locate(fb_ctx* ctx, int Q, int nx, int ny, const double* xs, const double* ys, int per_pair_grid, const double* vm,
                       int64_t K, const int32_t* pair_of, const double* pts, int32_t* tid, double* B);
int fb_pairs_strain(fb_ctx* ctx, fb_system* sys, int P, int nx, int ny, const double* xs, const double* ys, int64_t K, const int32_t* pid,
                    const double* xy0_fixed, const double* xy1_initial, const float* weight, const double* R, double stiffness_lambda,
                    double es0, int links_loaded, double default_strain, double* strain, int* iters, double* relres);
int fb_sys_lambda(fb_ctx* ctx, fb_system* sys, double stiffness_lambda, double crosslink_lambda, double* sl_out, double* cl_out);
int fb_sys_form(fb_ctx* ctx, fb_system* sys, double sl, double cl);
/* x: [2 nv] float64, x0 on entry when use_x0.  maxiter < 0: until converged, 0: zeros, > 0: cap */
int fb_sys_solve(fb_ctx* ctx, fb_system* sys, double* x, int use_x0, double rtol, double atol, int maxiter, int precond,
                 int* iters, double* relres);
int fb_sys_solve_fixed(fb_ctx* ctx, fb_system* sys, int iters, double* relres);
/* download: 0 K [nnzb][2][2] f64, 1 C [nnzb] f32 (nodal: C_xx = C_yy), 2 rhs [2nv] f64,
 * 3 stress [2nv] f32, 4 A [nnzb][2][2] f64, 5 b [2nv] f64 */
int fb_sys_get(fb_ctx* ctx, fb_system* sys, int which, void* out);

/* (fb_sys_solve: precond 2 = aggregation multigrid -- what the reference asks pyamg's smoothed_aggregation for,
 * optimizer.py:1962-1971, matcher.py:561: aggregates = grid cells of ~64 nodes per mesh, 3 rigid-body modes per aggregate,
 * Galerkin coarse operators, V(1,1) cycles with damped block Jacobi, coarsest level inverted directly; same fixed point as
 * precond 1, an order of magnitude fewer iterations on weakly pinned meshes.  Needs the meshes' stiffness assembled
 * (fb_sys_assemble_mesh*: the vertex coordinates of the aggregates are taken from there).
 * precond 3 = 'auto': precond 1 for at most the number of iterations a multigrid solve of this size costs in all (~30 ms; it
 * gives up earlier once the decay of its residual projects more than 1.5 x that), then precond 2 from the iterate reached; a hierarchy that cannot be built or stalls hands the iterate back to precond 1.)
 * optimizer.solve (optimizer.py:1945-2080) fixed point: Jacobi-preconditioned CG
 * on the symmetrised CSR system until ||Ax-b|| <= max(rtol, atol/||b||) ||b||.
 * x holds x0 on entry when use_x0 != 0.  precond: 0 none, 1 reference Jacobi
 * (optimizer.py:1962-1966). */
int fb_pcg(fb_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* idx, const double* val,
           const double* b, double* x, int use_x0, double rtol, double atol, int maxiter,
           int precond, int symmetrize, int* iters, double* relres);

/* resident form: upload once, iterate many times (bench / Newton steps) */
typedef struct fb_csr fb_csr;
int fb_csr_upload(fb_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* idx, const double* val,
                  int symmetrize, fb_csr** out);
void fb_csr_destroy(fb_ctx* ctx, fb_csr* A);
int fb_csr_info(fb_ctx* ctx, fb_csr* A, int64_t* n, int64_t* nnz, int64_t* nb, int64_t* nnzb);
int fb_spmv(fb_ctx* ctx, fb_csr* A, const double* x_host, double* y_host);
/* y = A x on device-resident vectors of length n (16-byte aligned), on the context's stream; no copies, no sync */
int fb_spmv_dev(fb_ctx* ctx, fb_csr* A, const double* x_dev, double* y_dev);
int fb_pcg_csr(fb_ctx* ctx, fb_csr* A, const double* b, double* x, int use_x0, double rtol, double atol,
               int maxiter, int precond, int* iters, double* relres);
/* vector side of the row-partitioned PCG of a coupled alignment window (aligner.py:510-535, 696-727 solve all free sections
 * of a window as one system; here its rows are partitioned by section over the ranks): Chronopoulos-Gear form, one fused
 * all-reduce of three scalars per iteration.  All pointers are device pointers, nothing synchronises the host.
 *   state double[8] = {gamma, alpha, beta, r.r, breakdown flag, iterations, -, -}
 *   fb_cgcg_update_dev : p = u + beta p; s = w + beta s; x += alpha p; r -= alpha s; u = minv r   (one pass over 12 vectors;
 *                        minv == NULL leaves u alone: the caller applies another preconditioner to r, e.g. the multigrid cycle)
 *   fb_cgcg_dots_dev   : out3 = (r.u, w.u, r.r) of the local rows, fixed summation order; scratch double[3 * 1024]
 *   fb_cgcg_scalars_dev: alpha, beta, gamma from the all-reduced out3 (first != 0: the start of the iteration); a
 *                        non-positive denominator zeroes the step and raises the flag instead of producing NaN */
int fb_cgcg_update_dev(fb_ctx* ctx, int64_t n, const double* state, const double* minv, double* x, double* r, double* u, const double* w,
                       double* p, double* s);
int fb_cgcg_dots_dev(fb_ctx* ctx, int64_t n, const double* r, const double* u, const double* w, double* scratch, double* out3);
int fb_cgcg_scalars_dev(fb_ctx* ctx, const double* t3, double* state, int first);
/* dst[i] = src[idx[i]], i < n, device pointers: the entries of u a neighbouring rank needs, packed for one transfer */
int fb_gather_f64_dev(fb_ctx* ctx, int64_t n, const int32_t* idx, const double* src, double* dst);

/* exactly `iters` PCG iterations with no convergence exit (throughput bench) */
int fb_pcg_fixed_iters(fb_ctx* ctx, fb_csr* A, const double* b_host, int iters, double* relres);

/* ------------------------------------------------------------------ exchange steps of a sharded run
 * (SURVEY.md sec.8b "fb_gatherv / fb_allgather (RCCL wrappers)", sec.8e).  The reference shards its pair / section
 * lists over worker PROCESSES and collects their results through the process pool (stitcher.py:375-392, 386-392;
 * aligner.py:588); here one process per GPU owns a contiguous shard and the results meet through RCCL over xGMI.
 * One communicator per context; every call enqueues on the context's stream and takes device pointers.  The 128-byte
 * id comes from fb_comm_unique_id on one rank and reaches the others out of band (torch.distributed store, a file, MPI).
 * librccl is bound at run time, so a single-GPU process never loads it; every failure returns FB_ERR_COMM. */
#define FB_COMM_ID_BYTES 128
#define FB_REDUCE_SUM 0
#define FB_REDUCE_MAX 1
typedef struct fb_comm fb_comm;
int fb_comm_unique_id(fb_ctx* ctx, void* id128);
int fb_comm_create(fb_ctx* ctx, const void* id128, int rank, int world, fb_comm** out);
void fb_comm_destroy(fb_ctx* ctx, fb_comm* comm);
int fb_comm_info(fb_ctx* ctx, fb_comm* comm, int* rank, int* world);
/* match table of every rank -> `root` (what Stitcher.dispatch_matchers merges from its workers, stitcher.py:386-392):
 * counts [world] = bytes of every rank's contribution (host array, the same on all ranks); recv (root only) gets the
 * contributions back to back in rank order.  Point-to-point transfers: nothing is padded, only the root receives. */
int fb_gatherv_dev(fb_ctx* ctx, fb_comm* comm, const void* send, const int64_t* counts, void* recv, int root);
/* equal contributions of bytes_per_rank (node displacements of a rank's sections; counts for fb_gatherv_dev) */
int fb_allgather_dev(fb_ctx* ctx, fb_comm* comm, const void* send, void* recv, size_t bytes_per_rank);
/* the fused scalar reduction of the coupled-window PCG (aligner.py:510-535, 696-727 solve one system over sections) */
int fb_allreduce_f64_dev(fb_ctx* ctx, fb_comm* comm, const double* send, double* recv, size_t n, int op);
/* halo exchange of the coupled window: grouped point-to-point transfers with the ranks of the neighbouring sections */
int fb_sendrecv_dev(fb_ctx* ctx, fb_comm* comm, int nsend, const int* send_peer, const void* const* send_ptr, const int64_t* send_bytes,
                    int nrecv, const int* recv_peer, void* const* recv_ptr, const int64_t* recv_bytes);
/* Coupled-window PCG (the fb_cgcg_* kernels above with these exchange steps between them).
   The whole loop behind one call: the rows of this rank as one square block-CSR `rows` of size n_loc + n_halo over [own | halo]
 * columns (halo rows empty), b / minv / x device double[n_loc] (x is overwritten, zero start).  Halo lists (host arrays): this
 * rank sends u[send_idx[send_off[k] .. send_off[k+1])] (send_idx: device int32, local row ids) to send_peer[k] and receives
 * halo entries [recv_off[k], recv_off[k+1]) from recv_peer[k]; comm may be NULL when there is nothing to exchange (one rank).
 * Per iteration: fb_cgcg_update_dev, fb_gather_f64_dev + fb_sendrecv_dev, fb_spmv_dev, fb_cgcg_dots_dev, ONE
 * fb_allreduce_f64_dev of 3 doubles, fb_cgcg_scalars_dev -- all on the context's stream; the host reads 64 bytes every
 * check_every iterations.  Stops at ||r|| <= rtol ||b|| (recurrence residual); maxiter < 0: no limit (1e5).  A dropped step
 * (p^T A p <= 0 in floating point) restarts the recurrence from the current iterate; check_every dropped steps in a row return
 * FB_ERR_BREAKDOWN, the iteration cap FB_ERR_NOCONV (x holds the last iterate).  bnorm: ||b|| over all ranks. */
int fb_cgcg_solve_dev(fb_ctx* ctx, fb_comm* comm, fb_csr* rows, int64_t n_loc, int64_t n_halo, const double* b, const double* minv, double* x,
                      int nsend, const int* send_peer, const int64_t* send_off, const int32_t* send_idx, int nrecv, const int* recv_peer,
                      const int64_t* recv_off, double rtol, int maxiter, int check_every, int* iters, double* relres, double* bnorm);

#ifdef __cplusplus
}
#endif
#endif /* FEABAS_HIP_H */
