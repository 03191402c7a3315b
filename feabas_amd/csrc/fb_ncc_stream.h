// Definitions shared by the streaming-class NCC translation units (fb_ncc.hip: launcher, generic and power-of-two kernels;
// fb_ncc_ct.hip: the compile-time mixed-radix kernels of the alignment block classes).
#pragma once
#include "fb_common.h"
#include "fb_ldsfft.h"

struct PeakPartial {
    float vmax;      // max of C in the chunk
    int imax;        // first index achieving it
    float mmax;      // max |Cm|
    int pad_;
    double sum;      // sum C      (STD confidence)
    double sumsq;    // sum C^2
};

// block descriptor of the crop loader: {image, x0, y0, h0, w0, x1, y1, h1, w1}
constexpr int kBlkStride = 9;

struct StreamGeom {
    int N, Fh, Fw, Sw, Kp, Hs, TR, TRI;   // Kp = ceil(Sw / 2): spectra are stored as interleaved column pairs
    int H0, W0, H1, W1;
    FftPlan pw, ph;
    const float2 *twW_hi, *twW_lo, *twH_hi, *twH_lo;
    const float* img0;
    const float* img1;
    const int* blk;
    int IH0, IW0, IH1, IW1;
    int want_q, want_std;
    const double* aff;            // per block affine gather of image 1 (crop mode only) or nullptr
    int xg_per8 = 0, xg_gx = 1, xg_gy = 1;   // > 0: a 1-D launch of 8 * xg_per8 workgroups over the (xg_gx, xg_gy) items, one contiguous eighth per XCD (fb_ncc_p2.inc: p2_item)
};

constexpr int kStreamThreads = 512;

__device__ __forceinline__ void peak_merge(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// ---- compile-time mixed-radix class (fb_ncc_ct.hip): lengths 2^a, 3 2^a, 5 2^a, 9 2^a with unrolled plans
bool fb_ncc_ct_len(int n);
// smallest compile-time length >= need that is at most 1.2 x ref (ref = the reference's next_fast_len(need)), or 0
int fb_ncc_ct_up(int need, int ref);
// rows per tile of the row / inverse-row kernels at row length Fw
int fb_ncc_ct_tr(int Fw);
// the three passes + sub-pixel neighbours of one sub-batch (T/V layouts of the generic kernels); g.TR, g.TRI, g.Hs set by the caller
int fb_ncc_ct_run(fb_ctx* ctx, const StreamGeom& g, int nb, float2* T0, float2* T1, float2* V0, float2* V1, PeakPartial* part,
                  int ntiles, float* ct9, int subpixel, double in_bytes);
