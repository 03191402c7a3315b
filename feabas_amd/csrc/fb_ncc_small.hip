// NCC path, on-chip class (FFT <= 256^2): placeholder dispatch until the fused LDS kernel lands.
#include "fb_common.h"

int fb_ncc_small_supported(int, int, int, int, int, int, int) { return 0; }

int fb_ncc_small_launch(fb_ctx* ctx, const float*, const float*, int, int, int, int, int, int, int, int, int, double*, double*, float*) {
    return fb_fail(ctx, FB_ERR_ARG, "ncc_small_fused not built");
}
