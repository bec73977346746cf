/* a17 -- multi-keyframe local BA (north-star extension; no reference counterpart). Placeholder launcher:
 * replaced by the FP64 Schur/MFMA implementation later in this round. */
#include "tb_internal.h"

size_t tbk_local_ba_work_bytes(int nkf, int nfixed, int npt, int nobs) { return 256; }

int tbk_local_ba(tb_ctx* ctx, const double K[4], int nkf, int nfixed, float* d_poses, int npt, float* d_pts,
                 const tb_ba_obs* d_obs, int nobs, int iters, double* d_stats, void* d_work, size_t work_bytes) {
    return tb_fail(ctx, TB_EUNSUPPORTED, "local BA kernel not built yet");
}
