/* Host side of libtb_hip.so: the C ABI of include/tb_capi.h on top of the HIP kernels (k_*.hip).
 * No CPU compute fallback lives here: every operator either runs its kernels or returns an error.
 * Host work is limited to set-up arithmetic the reference also does on the host (scale vectors, level
 * sizes, quotas, resize coefficient tables, cell tables), data movement, and the final ordering /
 * histogram bookkeeping of the window matcher.
 */
#include "tb_internal.h"
#include "tb_math.h"

#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <memory>

/* ------------------------------------------------------------------ errors / context */
int tb_fail(tb_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512]; /* no device binding here: formatting a message needs none, and TB_ENTER reports its own failure */
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int tb_scratch(tb_ctx* ctx, int slot, size_t bytes, void** out) { /* only called from entry points that have entered */
    if (bytes < 256) bytes = 256;
    if (ctx->scratch_cap[slot] < bytes) {
        if (ctx->scratch[slot]) {
            TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            TB_HIP(ctx, hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_cap[slot] = 0;
        }
        const size_t cap = bytes + bytes / 4;
        TB_HIP(ctx, hipMalloc(&ctx->scratch[slot], cap));
        ctx->scratch_cap[slot] = cap;
    }
    *out = ctx->scratch[slot];
    return TB_OK;
}

static hipEvent_t prof_event(tb_ctx* ctx) {
    if (!ctx->prof_pool.empty()) { hipEvent_t e = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
void tb_prof_begin(tb_ctx* ctx, const char* name) {
    ctx->prof_open = false;
    if (!ctx->prof) return;
    if (!ctx->prof_only.empty() && ctx->prof_only != name) return;
    ctx->prof_open = true;
    tb_ctx::ProfRec r;
    r.name = name;
    r.a = prof_event(ctx);
    r.b = prof_event(ctx);
    hipEventRecord(r.a, ctx->stream);
    ctx->prof_recs.push_back(r);
}
void tb_prof_end(tb_ctx* ctx) {
    if (!ctx->prof || !ctx->prof_open || ctx->prof_recs.empty()) return;
    hipEventRecord(ctx->prof_recs.back().b, ctx->stream);
    ctx->prof_open = false;
}
static void prof_drain(tb_ctx* ctx) {
    hipStreamSynchronize(ctx->stream);
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& acc = ctx->prof_acc[r.name];
            acc.first += 1;
            acc.second += ms;
        }
        ctx->prof_pool.push_back(r.a);
        ctx->prof_pool.push_back(r.b);
    }
    ctx->prof_recs.clear();
}

extern "C" {

int tb_profile_enable(tb_ctx* ctx, int on) {
    TB_ENTER(ctx);
    if (!ctx) return TB_EINVAL;
    prof_drain(ctx);
    ctx->prof_acc.clear();
    ctx->prof = on != 0;
    return TB_OK;
}

int tb_profile_only(tb_ctx* ctx, const char* kernel) {
    TB_ENTER(ctx);
    if (!ctx) return TB_EINVAL;
    ctx->prof_only = kernel ? kernel : "";
    return TB_OK;
}

int tb_profile_report(tb_ctx* ctx, char* buf, int cap) {
    TB_ENTER(ctx);
    if (!ctx || !buf || cap < 1) return TB_EINVAL;
    prof_drain(ctx);
    std::string out;
    char line[256];
    for (auto& kv : ctx->prof_acc) {
        snprintf(line, sizeof line, "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if ((int)out.size() + 1 > cap) return tb_fail(ctx, TB_ECAPACITY, "profile report needs %d bytes", (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return TB_OK;
}

const char* tb_version(void) { return "trackingbench-slam_amd 0.1 (gfx950)"; }

const char* tb_strerror(int code) {
    switch (code) {
        case TB_OK: return "ok";
        case TB_EINVAL: return "invalid argument";
        case TB_ENOMEM: return "out of memory";
        case TB_ECAPACITY: return "output capacity too small";
        case TB_EUNSUPPORTED: return "unsupported input (reference behaviour undefined)";
        case TB_EDEVICE: return "HIP device error";
        case TB_ESTATE: return "call sequence error";
        default: return "unknown error";
    }
}

int tb_create(int device, tb_ctx** out) {
    if (!out) return TB_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return TB_EDEVICE; /* no GPU: fail loudly */
    if (device < 0 || device >= ndev) return TB_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return TB_EDEVICE;
    tb_ctx* ctx = new (std::nothrow) tb_ctx();
    if (!ctx) return TB_ENOMEM;
    ctx->device = device;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) ctx->num_cu = ncu;
    }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return TB_EDEVICE;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return TB_OK;
}

int tb_measure_copy_seconds(tb_ctx* ctx, const void* d_src, void* d_dst, size_t bytes, int reps, double* seconds) {
    TB_ENTER(ctx);
    if (!ctx || !d_src || !d_dst || !seconds || reps < 1 || bytes < 16 || (bytes & 15) || ((uintptr_t)d_src & 15) || ((uintptr_t)d_dst & 15))
        return TB_EINVAL;
    hipEvent_t e0, e1;
    TB_HIP(ctx, hipEventCreate(&e0));
    TB_HIP(ctx, hipEventCreate(&e1));
    int rc = tbk_copy16(ctx, d_src, d_dst, bytes);
    if (rc == TB_OK) {
        hipError_t e = hipEventRecord(e0, ctx->stream);
        for (int i = 0; i < reps && rc == TB_OK; i++) rc = tbk_copy16(ctx, d_src, d_dst, bytes);
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess && rc == TB_OK) rc = tb_fail(ctx, TB_EDEVICE, "tb_measure_copy_seconds: %s", hipGetErrorString(e));
        *seconds = (double)ms * 1e-3 / reps;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

int tb_pack_rows_dev(tb_ctx* ctx, const void* src, int row_bytes, int cap, const int32_t* counts, int nframes, void* dst, long long* total) {
    TB_ENTER(ctx);
    if (!ctx || nframes < 0 || cap < 1 || row_bytes < 4 || (row_bytes & 3)) return TB_EINVAL;
    if (nframes == 0) return TB_OK;
    if (!src || !dst || !counts || src == dst) return TB_EINVAL;
    return tbk_pack_rows(ctx, src, row_bytes, cap, counts, nframes, dst, total);
}

int tb_set_concurrency(tb_ctx* ctx, int peers) {
    if (!ctx || peers < 1) return TB_EINVAL;
    ctx->peers = peers;
    return TB_OK;
}

int tb_debug_force_dense_fast(tb_ctx* ctx, int on) {
    if (!ctx) return TB_EINVAL;
    ctx->dbg_fast_dense = on ? 1 : 0;
    return TB_OK;
}

void tb_destroy(tb_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    ctx->plans.clear();
    while (!ctx->live.empty()) tb_extractor_destroy(*ctx->live.begin()); /* plans never outlive their context */
    for (int i = 0; i < 12; i++)
        if (ctx->scratch[i]) hipFree(ctx->scratch[i]);
    for (auto& g : ctx->ba_graphs) hipGraphExecDestroy(g.second);
    prof_drain(ctx);
    for (hipEvent_t e : ctx->prof_pool) hipEventDestroy(e);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* tb_last_error(const tb_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int tb_set_stream(tb_ctx* ctx, void* hip_stream) {
    TB_ENTER(ctx);
    if (!ctx) return TB_EINVAL;
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return TB_OK;
}

int tb_synchronize(tb_ctx* ctx) {
    TB_ENTER(ctx);
    if (!ctx) return TB_EINVAL;
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TB_OK;
}

/* ------------------------------------------------------------------ a1 / a2 / a3 host arithmetic */
int tb_scale_factors(int n, float scale, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2) {
    /* Frame::Frame, Frame.cpp:18-29 (float32 throughout) */
    if (n < 1 || !sf) return TB_EINVAL;
    float cur = 1.f, icur = 1.f;
    for (int i = 0; i < n; i++) {
        if (i > 0) { cur = cur * scale; icur = icur / scale; }
        sf[i] = cur;
        if (inv_sf) inv_sf[i] = icur;
        const float s2 = (i == 0) ? 1.f : cur * cur;
        if (sigma2) sigma2[i] = s2;
        if (inv_sigma2) inv_sigma2[i] = (i == 0) ? 1.f : 1.f / s2;
    }
    return TB_OK;
}

int tb_pyramid_sizes(int width, int height, int nlevels, const float* sf, int* widths, int* heights) {
    /* Frame::ComputePyramid, Frame.cpp:423-424: cv::Size(cols * scale, rows * scale) truncates */
    if (nlevels < 1 || !sf || !widths || !heights) return TB_EINVAL;
    widths[0] = width;
    heights[0] = height;
    for (int i = 1; i < nlevels; i++) {
        widths[i] = (int)((float)width * sf[i]);
        heights[i] = (int)((float)height * sf[i]);
    }
    return TB_OK;
}

int tb_orb_quotas(int nlevels, const float* sf, int target, int* quotas) {
    /* ORBExtractor::operator(), ORBextractor.cpp:919-930; reads sf[1], so one level is undefined there */
    if (nlevels < 2 || !sf || !quotas) return TB_EINVAL;
    float nDesired = target * (1 - sf[1]) / (1 - (float)pow((double)sf[1], (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        quotas[level] = tbm::cv_round(nDesired);
        sum += quotas[level];
        nDesired *= sf[1];
    }
    quotas[nlevels - 1] = std::max(target - sum, 0);
    return TB_OK;
}

/* ------------------------------------------------------------------ extractor plan */
static void build_resize_tables(int sw, int sh, int dw, int dh, std::vector<ResizeX>& rx, std::vector<ResizeY>& ry) {
    /* cv::resize INTER_LINEAR 8U coefficient set-up (OpenCV 3.3) */
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    rx.resize(dw);
    ry.resize(dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = tbm::cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        rx[dx].sx = (int16_t)sx;
        rx[dx].sx1 = (int16_t)std::min(sx + 1, sw - 1);
        rx[dx].a0 = (int16_t)tbm::cv_round((1.f - fx) * 2048.f);
        rx[dx].a1 = (int16_t)tbm::cv_round(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = tbm::cv_floor(fy);
        fy -= sy;
        ry[dy].sy0 = std::min(std::max(sy, 0), sh - 1);
        ry[dy].sy1 = std::min(std::max(sy + 1, 0), sh - 1);
        ry[dy].b0 = (int16_t)tbm::cv_round((1.f - fy) * 2048.f);
        ry[dy].b1 = (int16_t)tbm::cv_round(fy * 2048.f);
    }
}

static int fastgrid_ncell(int width, int height, int target) {
    const int cell = (int)sqrtf((float)width * (float)height / (float)target);
    if (cell < 1) return 0;
    const int cols = (int)((float)width / (float)cell), rows = (int)((float)height / (float)cell);
    return std::max((rows + 2) * (cols + 1), target);
}

void tb_extractor_destroy(tb_extractor* ex) {
    if (!ex) return;
    ex->ctx->live.erase(ex);
    for (auto it = ex->ctx->plans.begin(); it != ex->ctx->plans.end();)
        it = (it->second == ex) ? ex->ctx->plans.erase(it) : std::next(it);
    hipSetDevice(ex->ctx->device);
    hipStreamSynchronize(ex->ctx->stream);
    hipFree(ex->d_slab); hipFree(ex->d_img0_copy); hipFree(ex->d_blocks);
    for (int l = 0; l < TB_MAX_LEVELS; l++) { hipFree(ex->d_rx[l]); hipFree(ex->d_ry[l]); }
    hipFree(ex->d_cand); hipFree(ex->d_candCount); hipFree(ex->d_knode); hipFree(ex->d_sel); hipFree(ex->d_selCount);
    hipFree(ex->d_kps); hipFree(ex->d_desc); hipFree(ex->d_counts); hipFree(ex->d_exit); hipFree(ex->d_enode);
    hipFree(ex->d_gridBest); hipFree(ex->d_occ);
    delete ex;
}

int tb_extractor_create(tb_ctx* ctx, int width, int height, int nlevels, const float* sf, const int* widths,
                        const int* heights, int max_images, int max_target, tb_extractor** out) {
    TB_ENTER(ctx);
    if (!ctx || !out) return TB_EINVAL;
    *out = nullptr;
    if (width < 1 || height < 1 || nlevels < 1 || nlevels > TB_MAX_LEVELS || !sf || max_images < 1 || max_target < 1)
        return tb_fail(ctx, TB_EINVAL, "extractor_create: bad geometry %dx%d levels=%d images=%d target=%d", width, height,
                       nlevels, max_images, max_target);
    if (width > 4095 || height > 4095) return tb_fail(ctx, TB_EUNSUPPORTED, "images larger than 4095 px are not supported");
    TB_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<tb_extractor> exu(new tb_extractor());
    tb_extractor* ex = exu.get();
    ex->ctx = ctx;
    ex->max_images = max_images;
    ex->max_target = max_target;
    ex->sf.assign(sf, sf + nlevels);
    for (int l = 0; l < TB_MAX_LEVELS; l++) { ex->d_rx[l] = nullptr; ex->d_ry[l] = nullptr; ex->quotas[l] = 0; }
    PlanGeom& g = ex->g;
    memset(&g, 0, sizeof g);
    g.nlevels = nlevels;
    g.width = width;
    g.height = height;
    std::vector<int> ws(nlevels), hs(nlevels);
    if (widths && heights) {
        for (int l = 0; l < nlevels; l++) { ws[l] = widths[l]; hs[l] = heights[l]; }
        if (ws[0] != width || hs[0] != height) return tb_fail(ctx, TB_EINVAL, "level 0 size mismatch");
    } else {
        tb_pyramid_sizes(width, height, nlevels, sf, ws.data(), hs.data());
    }
    int maxq[TB_MAX_LEVELS] = {0};
    if (nlevels >= 2) tb_orb_quotas(nlevels, sf, max_target, maxq);
    std::vector<FastBlock> blocks;
    size_t off = 0, candOff = 0;
    int selBase = 0;
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& L = g.lv[l];
        if (ws[l] < 1 || hs[l] < 1 || ws[l] > 4095 || hs[l] > 4095)
            return tb_fail(ctx, TB_EINVAL, "level %d has size %dx%d", l, ws[l], hs[l]);
        L.w = ws[l];
        L.h = hs[l];
        L.stride = (L.w + 63) & ~63;
        L.off = off;
        off += ((size_t)L.stride * L.h + 255) & ~(size_t)255;
        L.sf = sf[l];
        L.inv_sf = 1.f;
        L.patchSize = (float)(int)(31 * sf[l]);
        /* ComputeKeyPointsOctTree grid, ORBextractor.cpp:749-763 */
        const int minB = TB_BORDER, maxBX = L.w - TB_BORDER, maxBY = L.h - TB_BORDER;
        const float fw = (float)(maxBX - minB), fh = (float)(maxBY - minB);
        L.nCols = (int)(fw / 30.f);
        L.nRows = (int)(fh / 30.f);
        L.cellBase = 0;
        L.nCells = 0;
        L.nIni = 0;
        L.hX = 1.f;
        L.wCell = L.hCell = 1;
        if (L.nCols >= 1 && L.nRows >= 1) {
            L.wCell = (int)ceilf(fw / (float)L.nCols);
            L.hCell = (int)ceilf(fh / (float)L.nRows);
            /* The reference walks the cells one by one (ORBextractor.cpp:765-786) and skips those that start at or
             * behind maxBorder - 3 (rows) / maxBorder - 6 (columns); what it does not skip but leaves without a scanned
             * pixel (an ROI under 7 px) yields nothing either. The cells that DO scan are a prefix in both directions,
             * and their scan regions tile [minB + 3, maxB - 3): count them, then cut the grid into blocks. */
            int nRowsEff = 0, nColsEff = 0;
            for (int i = 0; i < L.nRows; i++) {
                const float iniY = (float)minB + (float)i * (float)L.hCell;
                if (iniY >= (float)maxBY - 3.f) continue;
                if ((int)iniY + 3 < maxBY - 3) nRowsEff = i + 1;
            }
            for (int j = 0; j < L.nCols; j++) {
                const float iniX = minB + (float)(j * L.wCell);
                if (iniX >= (float)maxBX - 6.f) continue;
                if ((int)iniX + 3 < maxBX - 3) nColsEff = j + 1;
            }
            if (L.wCell + 6 + 15 > FB_S || L.hCell + 6 > FB_TH)
                return tb_fail(ctx, TB_EUNSUPPORTED, "level %d: FAST cell %dx%d exceeds the LDS tile", l, L.wCell, L.hCell);
            int bx = (FB_S - 6 - 15) / L.wCell, by = (FB_TH - 6) / L.hCell;
            bx = std::min(std::max(bx, 1), FB_MAX_CX);
            by = std::min(std::max(by, 1), FB_MAX_CY);
            for (int i0 = 0; i0 < nRowsEff; i0 += by)
                for (int j0 = 0; j0 < nColsEff; j0 += bx) {
                    FastBlock b;
                    b.level = (int16_t)l;
                    b.ncx = (int16_t)std::min(bx, nColsEff - j0);
                    b.ncy = (int16_t)std::min(by, nRowsEff - i0);
                    b.x0 = (int16_t)(minB + j0 * L.wCell);
                    b.y0 = (int16_t)(minB + i0 * L.hCell);
                    b.x1 = (int16_t)std::min(minB + (j0 + b.ncx) * L.wCell + 6, maxBX);
                    b.y1 = (int16_t)std::min(minB + (i0 + b.ncy) * L.hCell + 6, maxBY);
                    {   /* stage-1 lane map: the tile's column 0 is image column x0 & ~15 */
                        const int cx0 = b.x0 & 15, rw = b.x1 - b.x0, rh = b.y1 - b.y0;
                        const int scanX0 = cx0 + 3, scanX1 = cx0 + rw - 3;
                        b.sA = (int16_t)(scanX0 >> 4);
                        b.nss = (int16_t)(((scanX1 - 1) >> 4) - b.sA + 1);
                        b.rowsPer = (int16_t)(64 / b.nss);
                        b.nPass = (int16_t)((rh - 6 + b.rowsPer - 1) / b.rowsPer);
                        b.invNss = (uint16_t)((32768 + b.nss - 1) / b.nss);
                    }
                    blocks.push_back(b);
                }
            L.nCells = nRowsEff * nColsEff;
            /* DistributeOctTree, ORBextractor.cpp:498-500 (nIni < 1 clamped, see k_octree.hip) */
            int nIni = (int)roundf((float)(maxBX - minB) / (maxBY - minB));
            if (nIni < 1) nIni = 1;
            L.nIni = nIni;
            L.hX = (float)(maxBX - minB) / nIni;
        }
        L.candCap = ((L.w + 1) / 2) * ((L.h + 1) / 2) + 64;
        L.candOff = candOff;
        candOff += (size_t)L.candCap;
        L.quota = maxq[l];
        L.nodeCapAlloc = (L.nCells > 0) ? maxq[l] + 3 + 4 * L.nIni + 8 : 0;
        L.nodeCap = L.nodeCapAlloc;
        L.selBase = selBase;
        selBase += L.nodeCapAlloc;
    }
    g.slabBytes = off;
    g.candPerImage = candOff;
    g.selCap = std::max(std::max(selBase, fastgrid_ncell(width, height, max_target)), 64);
    ex->nBlocksTotal = (int)blocks.size();

    const size_t B = (size_t)max_images;
    TB_HIP(ctx, hipMalloc(&ex->d_slab, B * g.slabBytes));
    TB_HIP(ctx, hipMemsetAsync(ex->d_slab, 0, B * g.slabBytes, ctx->stream));
    if (!blocks.empty()) {
        TB_HIP(ctx, hipMalloc(&ex->d_blocks, blocks.size() * sizeof(FastBlock)));
        TB_HIP(ctx, hipMemcpy(ex->d_blocks, blocks.data(), blocks.size() * sizeof(FastBlock), hipMemcpyHostToDevice));
    }
    for (int l = 1; l < nlevels; l++) {
        std::vector<ResizeX> rx;
        std::vector<ResizeY> ry;
        build_resize_tables(ws[l - 1], hs[l - 1], ws[l], hs[l], rx, ry);
        TB_HIP(ctx, hipMalloc(&ex->d_rx[l], rx.size() * sizeof(ResizeX)));
        TB_HIP(ctx, hipMalloc(&ex->d_ry[l], ry.size() * sizeof(ResizeY)));
        TB_HIP(ctx, hipMemcpy(ex->d_rx[l], rx.data(), rx.size() * sizeof(ResizeX), hipMemcpyHostToDevice));
        TB_HIP(ctx, hipMemcpy(ex->d_ry[l], ry.data(), ry.size() * sizeof(ResizeY), hipMemcpyHostToDevice));
    }
    TB_HIP(ctx, hipMalloc(&ex->d_cand, B * g.candPerImage * sizeof(uint32_t)));
    TB_HIP(ctx, hipMalloc(&ex->d_knode, B * g.candPerImage * sizeof(uint32_t)));
    TB_HIP(ctx, hipMalloc(&ex->d_candCount, B * TB_MAX_LEVELS * sizeof(int32_t)));
    TB_HIP(ctx, hipMalloc(&ex->d_selCount, B * TB_MAX_LEVELS * sizeof(int32_t)));
    TB_HIP(ctx, hipMalloc(&ex->d_sel, B * g.selCap * sizeof(uint32_t)));
    TB_HIP(ctx, hipMalloc(&ex->d_kps, B * g.selCap * sizeof(tb_keypoint)));
    TB_HIP(ctx, hipMalloc(&ex->d_desc, B * g.selCap * 32));
    TB_HIP(ctx, hipMalloc(&ex->d_counts, B * sizeof(int32_t)));
    TB_HIP(ctx, hipMalloc(&ex->d_enode, 256));
    TB_HIP(ctx, hipMalloc(&ex->d_exit, 256));
    ex->enodeCap = 64;
    ex->exitCap = 32;
    TB_HIP(ctx, hipMemsetAsync(ex->d_counts, 0, B * sizeof(int32_t), ctx->stream));
    TB_HIP(ctx, hipMemsetAsync(ex->d_selCount, 0, B * TB_MAX_LEVELS * sizeof(int32_t), ctx->stream));
    TB_HIP(ctx, hipMemsetAsync(ex->d_candCount, 0, B * TB_MAX_LEVELS * sizeof(int32_t), ctx->stream));
    /* level 0 defaults to the slab until frames are attached */
    g.img0 = nullptr;
    g.img0_pitch = 0;
    g.img0_stride = 0;
    ctx->live.insert(ex);
    *out = exu.release();
    return TB_OK;
}

int tb_extractor_set_images_host(tb_extractor* ex, const uint8_t* images, int n, int stride, size_t pitch) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !images || n < 1 || n > ex->max_images || stride < ex->g.width) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    const LevelGeom& L0 = ex->g.lv[0];
    for (int b = 0; b < n; b++)
        TB_HIP(ctx, hipMemcpy2DAsync(ex->d_slab + (size_t)b * ex->g.slabBytes + L0.off, L0.stride, images + (size_t)b * pitch,
                                     stride, L0.w, L0.h, hipMemcpyHostToDevice, ctx->stream));
    ex->g.img0 = nullptr; /* level 0 lives in the slab */
    return TB_OK;
}

int tb_extractor_set_images_dev(tb_extractor* ex, const uint8_t* dev_images, int n, int stride, size_t pitch) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !dev_images || n < 1 || n > ex->max_images || stride < ex->g.width) return TB_EINVAL;
    ex->g.img0 = dev_images;
    ex->g.img0_stride = stride;
    ex->g.img0_pitch = pitch;
    return TB_OK;
}

int tb_extractor_set_levels_host(tb_extractor* ex, int index, const uint8_t* const* levels, const int* strides) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !levels || !strides || index < 0 || index >= ex->max_images) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    for (int l = 0; l < ex->g.nlevels; l++) {
        const LevelGeom& L = ex->g.lv[l];
        if (!levels[l] || strides[l] < L.w) return tb_fail(ctx, TB_EINVAL, "set_levels_host: level %d missing", l);
        TB_HIP(ctx, hipMemcpy2DAsync(ex->d_slab + (size_t)index * ex->g.slabBytes + L.off, L.stride, levels[l], strides[l], L.w,
                                     L.h, hipMemcpyHostToDevice, ctx->stream));
    }
    ex->g.img0 = nullptr;
    return TB_OK;
}

int tb_extractor_build_pyramid(tb_extractor* ex, int n) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || n < 1 || n > ex->max_images) return TB_EINVAL;
    for (int l = 1; l < ex->g.nlevels; l++) {
        int rc = tbk_resize_level(ex, l, n);
        if (rc) return rc;
    }
    return TB_OK;
}

int tb_extractor_get_level_host(tb_extractor* ex, int index, int level, uint8_t* out, int out_stride) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !out || index < 0 || index >= ex->max_images || level < 0 || level >= ex->g.nlevels) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    const LevelGeom& L = ex->g.lv[level];
    if (out_stride < L.w) return TB_EINVAL;
    const uint8_t* src;
    size_t sp;
    if (level == 0 && ex->g.img0) { src = ex->g.img0 + (size_t)index * ex->g.img0_pitch; sp = ex->g.img0_stride; }
    else { src = ex->d_slab + (size_t)index * ex->g.slabBytes + L.off; sp = L.stride; }
    TB_HIP(ctx, hipMemcpy2DAsync(out, out_stride, src, sp, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TB_OK;
}

int tb_extractor_orb(tb_extractor* ex, int n, int target, float init_th, float min_th, int quota_mode,
                     const tb_keypoint* exit_keys, int n_exit) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || n < 1 || n > ex->max_images || target < 0 || n_exit < 0 || (n_exit > 0 && !exit_keys)) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    PlanGeom& g = ex->g;
    if (quota_mode == 0) {
        if (g.nlevels < 2) return tb_fail(ctx, TB_EUNSUPPORTED, "ORB extraction reads sf[1]: needs >= 2 levels");
        if (target > ex->max_target) return tb_fail(ctx, TB_ECAPACITY, "target %d exceeds plan max_target %d", target, ex->max_target);
        tb_orb_quotas(g.nlevels, ex->sf.data(), target, ex->quotas);
        ex->have_quotas = true;
    } else if (!ex->have_quotas) {
        /* AddPoints before operator(): the reference indexes an empty mnFeaturesPerLevel (ORBextractor.cpp:810) */
        return tb_fail(ctx, TB_ESTATE, "AddPoints-mode extraction before any operator()-mode call");
    }
    for (int l = 0; l < g.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        L.quota = ex->quotas[l];
        L.nodeCap = (L.nCells > 0) ? L.quota + 3 + 4 * L.nIni : 0;
        if (L.nodeCap > L.nodeCapAlloc) return tb_fail(ctx, TB_ECAPACITY, "level %d quota %d exceeds the plan", l, L.quota);
    }
    /* cv::FAST clamps its threshold to [0,255]; (int) truncation as at ORBextractor.cpp:786,791 */
    const int ith = std::min(std::max((int)init_th, 0), 255), mth = std::min(std::max((int)min_th, 0), 255);
    if (n_exit > 0) {
        if (n_exit > ex->exitCap) {
            TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ex->d_exit);
            ex->d_exit = nullptr;
            TB_HIP(ctx, hipMalloc(&ex->d_exit, (size_t)n_exit * 2 * sizeof(float)));
            ex->exitCap = n_exit;
        }
        const size_t need = (size_t)n * g.nlevels * n_exit;
        if (need > ex->enodeCap) {
            TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ex->d_enode);
            ex->d_enode = nullptr;
            TB_HIP(ctx, hipMalloc(&ex->d_enode, need * sizeof(int32_t)));
            ex->enodeCap = need;
        }
        std::vector<float> xy((size_t)n_exit * 2);
        for (int i = 0; i < n_exit; i++) { xy[2 * i] = exit_keys[i].x; xy[2 * i + 1] = exit_keys[i].y; }
        TB_HIP(ctx, hipMemcpyAsync(ex->d_exit, xy.data(), xy.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        TB_HIP(ctx, hipStreamSynchronize(ctx->stream)); /* xy is a stack-lifetime staging buffer */
    }
    int rc = tbk_fast_cells(ex, n, ith, mth);
    if (rc) return rc;
    rc = tbk_octree(ex, n, n_exit);
    if (rc) return rc;
    rc = tbk_describe(ex, n);
    if (rc) return rc;
    ex->last_n = n;
    ex->last_was_orb = true;
    return TB_OK;
}

int tb_extractor_fastgrid(tb_extractor* ex, int n, const float* inv_sf, int target, float threshold,
                          const uint8_t* occupancy, int n_occupancy) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || n < 1 || n > ex->max_images || !inv_sf || target < 1) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    for (int l = 0; l < ex->g.nlevels; l++) ex->g.lv[l].inv_sf = inv_sf[l];
    int n_occ = 0;
    if (occupancy && n_occupancy > 0) {
        if ((size_t)n_occupancy > ex->occCap) {
            TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ex->d_occ);
            ex->d_occ = nullptr;
            TB_HIP(ctx, hipMalloc(&ex->d_occ, (size_t)n_occupancy));
            ex->occCap = (size_t)n_occupancy;
        }
        TB_HIP(ctx, hipMemcpyAsync(ex->d_occ, occupancy, (size_t)n_occupancy, hipMemcpyHostToDevice, ctx->stream));
        TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        n_occ = n_occupancy;
    }
    int rc = tbk_fastgrid(ex, n, target, threshold, n_occ);
    if (rc) return rc;
    ex->last_n = n;
    ex->last_was_orb = false;
    return TB_OK;
}

int tb_extractor_counts_host(tb_extractor* ex, int n, int* counts) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !counts || n < 1 || n > ex->max_images) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    TB_HIP(ctx, hipMemcpyAsync(counts, ex->d_counts, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TB_OK;
}

int tb_extractor_results_host(tb_extractor* ex, int index, tb_keypoint* kps, uint8_t* desc, int cap, int* count) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !count || index < 0 || index >= ex->max_images) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    int32_t c = 0;
    TB_HIP(ctx, hipMemcpyAsync(&c, ex->d_counts + index, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = c;
    if (c > cap) return tb_fail(ctx, TB_ECAPACITY, "results: %d keypoints, capacity %d", c, cap);
    if (c > 0 && kps)
        TB_HIP(ctx, hipMemcpyAsync(kps, ex->d_kps + (size_t)index * ex->g.selCap, (size_t)c * sizeof(tb_keypoint),
                                   hipMemcpyDeviceToHost, ctx->stream));
    if (c > 0 && desc && ex->last_was_orb)
        TB_HIP(ctx, hipMemcpyAsync(desc, ex->d_desc + (size_t)index * ex->g.selCap * 32, (size_t)c * 32, hipMemcpyDeviceToHost,
                                   ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TB_OK;
}

int tb_extractor_results_dev(tb_extractor* ex, const tb_keypoint** kps, const uint8_t** desc, const int32_t** counts,
                             int* kp_capacity) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex) return TB_EINVAL;
    if (kps) *kps = ex->d_kps;
    if (desc) *desc = ex->d_desc;
    if (counts) *counts = ex->d_counts;
    if (kp_capacity) *kp_capacity = ex->g.selCap;
    return TB_OK;
}

__global__ void k_copy_results(const tb_keypoint* __restrict__ skp, const uint8_t* __restrict__ sdesc,
                               const int32_t* __restrict__ scnt, int selCap, tb_keypoint* __restrict__ dkp,
                               uint8_t* __restrict__ ddesc, int32_t* __restrict__ dcnt, int cap) {
    const int b = blockIdx.y;
    const int c = min(scnt[b], cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) dcnt[b] = c;
    /* 60 bytes per keypoint row: 7 dwords of tb_keypoint + 8 dwords of descriptor */
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c * 15) return;
    const int row = i / 15, w = i - row * 15;
    if (w < 7) reinterpret_cast<uint32_t*>(dkp + (size_t)b * cap + row)[w] = reinterpret_cast<const uint32_t*>(skp + (size_t)b * selCap + row)[w];
    else reinterpret_cast<uint32_t*>(ddesc + ((size_t)b * cap + row) * 32)[w - 7] =
             reinterpret_cast<const uint32_t*>(sdesc + ((size_t)b * selCap + row) * 32)[w - 7];
}

int tb_extractor_copy_results_dev(tb_extractor* ex, int n, tb_keypoint* kps, uint8_t* desc, int32_t* counts, int cap) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || n < 1 || n > ex->max_images || !kps || !desc || !counts || cap < 1) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    const int rows = std::min(cap, ex->g.selCap);
    tb_prof_begin(ctx, "k_copy_results");
    hipLaunchKernelGGL(k_copy_results, dim3((rows * 15 + 255) / 256, n), dim3(256), 0, ctx->stream, ex->d_kps, ex->d_desc,
                       ex->d_counts, ex->g.selCap, kps, desc, counts, cap);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}

int tb_extractor_candidates_host(tb_extractor* ex, int index, int level, tb_corner* out, int cap, int* count) {
    TB_ENTER((ex ? ex->ctx : nullptr));
    if (!ex || !count || index < 0 || index >= ex->max_images || level < 0 || level >= ex->g.nlevels) return TB_EINVAL;
    tb_ctx* ctx = ex->ctx;
    const LevelGeom& L = ex->g.lv[level];
    int32_t c = 0;
    TB_HIP(ctx, hipMemcpyAsync(&c, ex->d_candCount + index * TB_MAX_LEVELS + level, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (c > L.candCap) return tb_fail(ctx, TB_ECAPACITY, "candidate overflow on level %d", level);
    *count = c;
    if (c > cap) return tb_fail(ctx, TB_ECAPACITY, "candidates: %d, capacity %d", c, cap);
    std::vector<uint32_t> rec((size_t)c);
    if (c > 0) {
        TB_HIP(ctx, hipMemcpyAsync(rec.data(), ex->d_cand + (size_t)index * ex->g.candPerImage + L.candOff, (size_t)c * 4,
                                   hipMemcpyDeviceToHost, ctx->stream));
        TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    /* the kernel emits in arrival order; present them in the reference's order (cell-major raster) */
    std::vector<std::pair<uint64_t, uint32_t>> keyed((size_t)c);
    for (int i = 0; i < c; i++) {
        const int x = rec[i] & 0xfff, y = (rec[i] >> 12) & 0xfff;
        const int ci = (y - 3) / L.hCell, cj = (x - 3) / L.wCell;
        keyed[i] = std::make_pair(((uint64_t)(ci * L.nCols + cj) << 24) | ((uint64_t)y << 12) | (uint64_t)x, rec[i]);
    }
    std::sort(keyed.begin(), keyed.end());
    for (int i = 0; i < c; i++) {
        out[i].x = keyed[i].second & 0xfff;
        out[i].y = (keyed[i].second >> 12) & 0xfff;
        out[i].score = keyed[i].second >> 24;
    }
    return TB_OK;
}

/* ------------------------------------------------------------------ single-frame operator forms */
static int get_plan(tb_ctx* ctx, const char* tag, int nlevels, const float* sf, const int* ws, const int* hs, int max_target,
                    tb_extractor** out) {
    std::string key = tag;
    char buf[64];
    for (int l = 0; l < nlevels; l++) {
        snprintf(buf, sizeof buf, ":%dx%d:%08x", ws[l], hs[l], *reinterpret_cast<const uint32_t*>(&sf[l]));
        key += buf;
    }
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end() && it->second->max_target >= max_target) { *out = it->second; return TB_OK; }
    if (it != ctx->plans.end()) tb_extractor_destroy(it->second); /* also drops the cache entry */
    tb_extractor* ex = nullptr;
    int rc = tb_extractor_create(ctx, ws[0], hs[0], nlevels, sf, ws, hs, 1, std::max(max_target, 2048), &ex);
    if (rc) return rc;
    ctx->plans[key] = ex;
    *out = ex;
    return TB_OK;
}

int tb_pyramid(tb_ctx* ctx, const uint8_t* image, int width, int height, int stride, int nlevels, const float* sf,
               uint8_t* const* levels_out, const int* strides_out) {
    TB_ENTER(ctx);
    if (!ctx || !image || !sf || !levels_out || !strides_out || nlevels < 1 || nlevels > TB_MAX_LEVELS) return TB_EINVAL;
    std::vector<int> ws(nlevels), hs(nlevels);
    tb_pyramid_sizes(width, height, nlevels, sf, ws.data(), hs.data());
    tb_extractor* ex = nullptr;
    int rc = get_plan(ctx, "pyr", nlevels, sf, ws.data(), hs.data(), 1, &ex);
    if (rc) return rc;
    rc = tb_extractor_set_images_host(ex, image, 1, stride, 0);
    if (rc) return rc;
    rc = tb_extractor_build_pyramid(ex, 1);
    if (rc) return rc;
    for (int l = 1; l < nlevels; l++) {
        if (!levels_out[l]) continue;
        rc = tb_extractor_get_level_host(ex, 0, l, levels_out[l], strides_out[l]);
        if (rc) return rc;
    }
    return tb_synchronize(ctx);
}

int tb_fast_detect(tb_ctx* ctx, const uint8_t* image, int width, int height, int stride, int threshold, int nms,
                   tb_corner* out, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !image || !count || width < 0 || height < 0 || width > 4095 || height > 4095) return TB_EINVAL;
    *count = 0;
    if (width < 7 || height < 7) return TB_OK;
    threshold = std::min(std::max(threshold, 0), 255);
    void *d_img, *d_out, *d_cnt;
    const int rcap = (nms ? ((width + 1) / 2) * ((height + 1) / 2) : width * height) + 64;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)width * height, &d_img))) return rc;
    if ((rc = tb_scratch(ctx, 1, (size_t)rcap * 4, &d_out))) return rc;
    if ((rc = tb_scratch(ctx, 2, 256, &d_cnt))) return rc;
    TB_HIP(ctx, hipMemcpy2DAsync(d_img, width, image, stride, width, height, hipMemcpyHostToDevice, ctx->stream));
    rc = tbk_fast_image(ctx, (const uint8_t*)d_img, width, height, width, threshold, nms, 9, (uint32_t*)d_out, rcap, (int32_t*)d_cnt);
    if (rc) return rc;
    int32_t c = 0;
    TB_HIP(ctx, hipMemcpyAsync(&c, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = c;
    if (c > cap || c > rcap) return tb_fail(ctx, TB_ECAPACITY, "fast_detect: %d corners, capacity %d", c, cap);
    std::vector<uint32_t> rec((size_t)c);
    if (c > 0) {
        TB_HIP(ctx, hipMemcpy(rec.data(), d_out, (size_t)c * 4, hipMemcpyDeviceToHost));
        /* raster order of cv::FAST: sort by (y, x) */
        std::sort(rec.begin(), rec.end(), [](uint32_t a, uint32_t b) { return (a & 0xffffff) < (b & 0xffffff); });
    }
    for (int i = 0; i < c && out; i++) {
        out[i].x = rec[i] & 0xfff;
        out[i].y = (rec[i] >> 12) & 0xfff;
        out[i].score = rec[i] >> 24;
    }
    return TB_OK;
}

int tb_orb_extract(tb_ctx* ctx, const uint8_t* const* levels, const int* widths, const int* heights, const int* strides,
                   int nlevels, const float* sf, int target, float init_th, float min_th, const tb_keypoint* exit_keys,
                   int n_exit, int use_quotas, int* quotas_inout, tb_keypoint* kps, uint8_t* desc, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !levels || !widths || !heights || !strides || !sf || !quotas_inout || !count || nlevels < 1 ||
        nlevels > TB_MAX_LEVELS)
        return TB_EINVAL;
    *count = 0;
    if (!levels[0] || widths[0] < 1 || heights[0] < 1) return TB_OK; /* images.at(0).empty(): silent return */
    if (nlevels < 2) return tb_fail(ctx, TB_EUNSUPPORTED, "ORB extraction reads sf[1]: needs >= 2 levels");
    int qsum = 0;
    if (use_quotas) for (int l = 0; l < nlevels; l++) qsum += quotas_inout[l];
    tb_extractor* ex = nullptr;
    int rc = get_plan(ctx, "orb", nlevels, sf, widths, heights, std::max(target, qsum), &ex);
    if (rc) return rc;
    rc = tb_extractor_set_levels_host(ex, 0, levels, strides);
    if (rc) return rc;
    if (use_quotas) {
        for (int l = 0; l < nlevels; l++) ex->quotas[l] = quotas_inout[l];
        ex->have_quotas = true;
    }
    rc = tb_extractor_orb(ex, 1, target, init_th, min_th, use_quotas ? 1 : 0, exit_keys, n_exit);
    if (rc) return rc;
    if (!use_quotas) for (int l = 0; l < nlevels; l++) quotas_inout[l] = ex->quotas[l];
    return tb_extractor_results_host(ex, 0, kps, desc, cap, count);
}

int tb_fastgrid_extract(tb_ctx* ctx, const uint8_t* const* levels, const int* widths, const int* heights, const int* strides,
                        int nlevels, const float* inv_sf, int target, float threshold, const uint8_t* occupancy,
                        int n_occupancy, tb_keypoint* kps, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !levels || !widths || !heights || !strides || !inv_sf || !count || nlevels < 1 || nlevels > TB_MAX_LEVELS ||
        target < 1)
        return TB_EINVAL;
    *count = 0;
    if (!levels[0] || widths[0] < 1 || heights[0] < 1) return TB_OK;
    std::vector<float> sf(nlevels);
    for (int l = 0; l < nlevels; l++) sf[l] = 1.f / inv_sf[l]; /* plan key + ORB fields only; unused by fastgrid */
    tb_extractor* ex = nullptr;
    int rc = get_plan(ctx, "fg", nlevels, sf.data(), widths, heights, target, &ex);
    if (rc) return rc;
    rc = tb_extractor_set_levels_host(ex, 0, levels, strides);
    if (rc) return rc;
    rc = tb_extractor_fastgrid(ex, 1, inv_sf, target, threshold, occupancy, n_occupancy);
    if (rc) return rc;
    return tb_extractor_results_host(ex, 0, kps, nullptr, cap, count);
}

/* ------------------------------------------------------------------ matchers */
int tb_descriptor_distance(const uint8_t* a, const uint8_t* b) {
    /* Matcher::DescriptorDistance, matcher.cpp:793-808: 256-bit Hamming distance */
    int dist = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        dist += __builtin_popcountll(x ^ y);
    }
    return dist;
}

void tb_three_maxima(const int* sizes, int L, int* ind1, int* ind2, int* ind3) {
    /* Matcher::ComputeThreeMaxima, matcher.cpp:810-851 (caller initialises the indices, :379) */
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

static int bf_host(tb_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int crosscheck, int filter, float ratio,
                   float min_th, tb_match* out, int cap, int* count) {
    if (!ctx || !count || n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2)) return TB_EINVAL;
    *count = 0;
    if (n1 == 0 || n2 == 0) return TB_OK;
    const int max_n = std::max(n1, n2);
    const size_t pitch = (size_t)max_n * 32;
    void *dd1, *dd2, *tb, *qb, *dout, *dcnt;
    int rc;
    if ((rc = tb_scratch(ctx, 0, pitch, &dd1))) return rc;
    if ((rc = tb_scratch(ctx, 1, pitch, &dd2))) return rc;
    if ((rc = tb_scratch(ctx, 2, (size_t)max_n * 8, &tb))) return rc;
    if ((rc = tb_scratch(ctx, 3, (size_t)max_n * 8, &qb))) return rc;
    if ((rc = tb_scratch(ctx, 4, (size_t)n1 * sizeof(tb_match), &dout))) return rc;
    if ((rc = tb_scratch(ctx, 5, 256, &dcnt))) return rc;
    int32_t cnts[3] = {n1, n2, 0};
    TB_HIP(ctx, hipMemcpyAsync(dd1, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dd2, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dcnt, cnts, sizeof cnts, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = tbk_bf_batch(ctx, 1, (const uint8_t*)dd1, (const int32_t*)dcnt, (const uint8_t*)dd2, (const int32_t*)dcnt + 1, pitch,
                      max_n, crosscheck, filter, ratio, min_th, (tb_match*)dout, n1, (int32_t*)dcnt + 2,
                      (unsigned long long*)tb, (unsigned long long*)qb);
    if (rc) return rc;
    int32_t c = 0;
    TB_HIP(ctx, hipMemcpyAsync(&c, (int32_t*)dcnt + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = c;
    if (c > cap) return tb_fail(ctx, TB_ECAPACITY, "matches: %d, capacity %d", c, cap);
    if (c > 0 && out) TB_HIP(ctx, hipMemcpy(out, dout, (size_t)c * sizeof(tb_match), hipMemcpyDeviceToHost));
    return TB_OK;
}

int tb_match_bf(tb_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, int crosscheck, tb_match* out, int cap,
                int* count) {
    TB_ENTER(ctx);
    return bf_host(ctx, d1, n1, d2, n2, crosscheck, 0, 0.f, 0.f, out, cap, count);
}

int tb_search_by_bf(tb_ctx* ctx, const uint8_t* d1, int n1, const uint8_t* d2, int n2, float ratio, float min_th,
                    tb_match* out, int cap, int* count) {
    TB_ENTER(ctx);
    return bf_host(ctx, d1, n1, d2, n2, 1, 1, ratio, min_th, out, cap, count);
}

int tb_search_by_bf_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* desc1, const int32_t* counts1, const uint8_t* desc2,
                              const int32_t* counts2, size_t set_pitch, float ratio, float min_th, tb_match* out, int cap,
                              int32_t* out_counts) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || !desc1 || !desc2 || !counts1 || !counts2 || !out || !out_counts || set_pitch < 32 || cap < 1)
        return TB_EINVAL;
    if (npairs == 0) return TB_OK;
    const int max_n = (int)(set_pitch / 32);
    void *tb, *qb;
    int rc;
    if ((rc = tb_scratch(ctx, 2, (size_t)npairs * max_n * 8, &tb))) return rc;
    if ((rc = tb_scratch(ctx, 3, (size_t)npairs * max_n * 8, &qb))) return rc;
    return tbk_bf_batch(ctx, npairs, desc1, counts1, desc2, counts2, set_pitch, max_n, 1, 1, ratio, min_th, out, cap, out_counts,
                        (unsigned long long*)tb, (unsigned long long*)qb);
}

/* ---- DBoW2 transform (see include/tb_capi.h) */
struct tb_vocab {
    tb_ctx* ctx = nullptr;
    int nnodes = 0, k = 0, L = 0, weighting = 0, scoring = 0;
    int32_t *d_child_start = nullptr, *d_child_items = nullptr, *d_word_id = nullptr;
    uint8_t* d_desc = nullptr;
    double* d_weight = nullptr;
};

void tb_vocab_destroy(tb_vocab* v) {
    if (!v) return;
    if (v->ctx) hipSetDevice(v->ctx->device);
    hipFree(v->d_child_start); hipFree(v->d_child_items); hipFree(v->d_word_id); hipFree(v->d_desc); hipFree(v->d_weight);
    delete v;
}

int tb_vocab_create(tb_ctx* ctx, const tb_vocabulary* h, tb_vocab** out) {
    TB_ENTER(ctx);
    if (!ctx || !h || !out || h->nnodes < 1 || !h->child_start || !h->desc || !h->word_id || !h->weight) return TB_EINVAL;
    *out = nullptr;
    const int nn = h->nnodes, nc = h->child_start[nn];
    /* the tree must be walkable without a bounds test in the kernel: offsets ascending, children in range, no node its own
     * ancestor (children have larger ids than their parents in every DBoW2 file: ids are assigned in creation order) */
    if (h->child_start[0] != 0 || nc < 0 || nc > nn || (nc && !h->child_items)) return tb_fail(ctx, TB_EINVAL, "vocabulary: child offsets");
    for (int n = 0; n < nn; n++) {
        if (h->child_start[n + 1] < h->child_start[n]) return tb_fail(ctx, TB_EINVAL, "vocabulary: child offsets of node %d", n);
        for (int c = h->child_start[n]; c < h->child_start[n + 1]; c++)
            if (h->child_items[c] <= n || h->child_items[c] >= nn) return tb_fail(ctx, TB_EINVAL, "vocabulary: child %d of node %d", h->child_items[c], n);
    }
    tb_vocab* v = new (std::nothrow) tb_vocab();
    if (!v) return TB_ENOMEM;
    v->ctx = ctx; v->nnodes = nn; v->k = h->k; v->L = h->L; v->weighting = h->weighting; v->scoring = h->scoring;
    hipError_t e = hipMalloc(&v->d_child_start, (size_t)(nn + 1) * 4);
    if (e == hipSuccess) e = hipMalloc(&v->d_child_items, (size_t)std::max(nc, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(&v->d_word_id, (size_t)nn * 4);
    if (e == hipSuccess) e = hipMalloc(&v->d_desc, (size_t)nn * 32);
    if (e == hipSuccess) e = hipMalloc(&v->d_weight, (size_t)nn * 8);
    if (e == hipSuccess) e = hipMemcpy(v->d_child_start, h->child_start, (size_t)(nn + 1) * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && nc) e = hipMemcpy(v->d_child_items, h->child_items, (size_t)nc * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_word_id, h->word_id, (size_t)nn * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_desc, h->desc, (size_t)nn * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_weight, h->weight, (size_t)nn * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) { tb_vocab_destroy(v); return tb_fail(ctx, TB_EDEVICE, "vocabulary upload: %s", hipGetErrorString(e)); }
    *out = v;
    return TB_OK;
}

int tb_bow_transform_batch_dev(tb_ctx* ctx, const tb_vocab* voc, int nframes, const uint8_t* desc, const int32_t* counts,
                               int desc_pitch, int levelsup, int32_t* word_ids, int32_t* node_ids, double* weights,
                               uint64_t* fv_keys, int32_t* fv_counts) {
    TB_ENTER(ctx);
    if (!ctx || !voc || voc->ctx != ctx || nframes < 0 || desc_pitch < 0 || levelsup < 0) return TB_EINVAL;
    if (nframes == 0 || desc_pitch == 0) return TB_OK;
    if (!desc || (fv_keys && (!fv_counts || desc_pitch > 8192))) return TB_EINVAL;
    void *dn = node_ids, *dwt = weights;
    int rc;
    if (fv_keys && !node_ids && (rc = tb_scratch(ctx, 4, (size_t)nframes * desc_pitch * 4, &dn))) return rc;
    if (fv_keys && !weights && (rc = tb_scratch(ctx, 5, (size_t)nframes * desc_pitch * 8, &dwt))) return rc;
    return tbk_bow_transform(ctx, voc->nnodes, voc->L, voc->d_child_start, voc->d_child_items, voc->d_desc, voc->d_word_id, voc->d_weight,
                             nframes, desc, counts, desc_pitch, levelsup, word_ids, (int32_t*)dn, (double*)dwt,
                             (unsigned long long*)fv_keys, fv_counts);
}

int tb_bow_transform(tb_ctx* ctx, const tb_vocab* voc, const uint8_t* desc, int n, int levelsup, int32_t* word_ids, double* weights,
                     int32_t* node_ids) {
    TB_ENTER(ctx);
    if (!ctx || !voc || voc->ctx != ctx || n < 0 || levelsup < 0 || (n && (!desc || !word_ids || !weights || !node_ids))) return TB_EINVAL;
    if (n == 0) return TB_OK;
    void *dd, *dw, *dn, *dwt;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)n * 32, &dd)) || (rc = tb_scratch(ctx, 1, (size_t)n * 4, &dw)) ||
        (rc = tb_scratch(ctx, 2, (size_t)n * 4, &dn)) || (rc = tb_scratch(ctx, 3, (size_t)n * 8, &dwt)))
        return rc;
    hipStream_t s = ctx->stream;
    TB_HIP(ctx, hipMemcpyAsync(dd, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
    if ((rc = tbk_bow_transform(ctx, voc->nnodes, voc->L, voc->d_child_start, voc->d_child_items, voc->d_desc, voc->d_word_id, voc->d_weight,
                                1, (const uint8_t*)dd, nullptr, n, levelsup, (int32_t*)dw, (int32_t*)dn, (double*)dwt, nullptr, nullptr)))
        return rc;
    TB_HIP(ctx, hipMemcpyAsync(word_ids, dw, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipMemcpyAsync(node_ids, dn, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipMemcpyAsync(weights, dwt, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipStreamSynchronize(s));
    return TB_OK;
}

int tb_search_by_bow_batch_dev(tb_ctx* ctx, int npairs, const tb_keypoint* k1, const uint8_t* d1, int pitch1, const uint64_t* fv1,
                               const int32_t* fv_counts1, const tb_keypoint* k2, const uint8_t* d2, int pitch2, const uint64_t* fv2,
                               const int32_t* fv_counts2, const uint8_t* has_mp2, int map_point_only, int th_low, float nratio,
                               int histo_len, int check_orientation, tb_match* out, int cap, int32_t* out_counts, int32_t* flags) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || pitch1 < 1 || pitch2 < 1 || histo_len < 1 || histo_len > 1024 || cap < 0) return TB_EINVAL;
    if (npairs == 0) return TB_OK;
    if (!k1 || !d1 || !fv1 || !fv_counts1 || !k2 || !d2 || !fv2 || !fv_counts2 || !out_counts || !flags || (cap && !out)) return TB_EINVAL;
    void* best;
    int rc;
    if ((rc = tb_scratch(ctx, 6, (size_t)npairs * pitch1 * 16, &best))) return rc;
    return tbk_bow_search_batch(ctx, npairs, k1, d1, pitch1, (const unsigned long long*)fv1, fv_counts1, k2, d2, pitch2,
                                (const unsigned long long*)fv2, fv_counts2, has_mp2, map_point_only, th_low, nratio, histo_len,
                                check_orientation, out, cap, out_counts, flags, (int32_t*)best);
}

int tb_stereo_tracks_to_obs_batch_dev(tb_ctx* ctx, int nframes, const tb_keypoint* keys_left, const tb_keypoint* keys_right,
                                      int key_pitch, const tb_match* matches, const int32_t* match_counts, int match_pitch,
                                      const float K[4], float bf, const float* inv_sigma2, int nlevels, tb_obs* obs, int obs_pitch,
                                      int32_t* obs_counts) {
    TB_ENTER(ctx);
    if (!ctx || nframes < 0 || !K || !inv_sigma2 || nlevels < 1 || nlevels > TB_MAX_LEVELS || key_pitch < 1 || match_pitch < 1 || obs_pitch < 1)
        return TB_EINVAL;
    if (nframes == 0) return TB_OK;
    if (!keys_left || !keys_right || !matches || !match_counts || !obs || !obs_counts) return TB_EINVAL;
    void* dsig;
    int rc;
    if ((rc = tb_scratch(ctx, 4, TB_MAX_LEVELS * sizeof(float), &dsig))) return rc;
    /* the table is a few floats of host memory: staged through a pinned-free async copy (the stream orders it before the kernel) */
    TB_HIP(ctx, hipMemcpyAsync(dsig, inv_sigma2, (size_t)nlevels * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return tbk_stereo_obs(ctx, nframes, keys_left, keys_right, key_pitch, matches, match_counts, match_pitch, K, bf, (const float*)dsig, nlevels,
                          obs, obs_pitch, obs_counts);
}

int tb_search_by_violence(tb_ctx* ctx, const tb_keypoint* k1, const uint8_t* d1, int n1, const tb_keypoint* k2,
                          const uint8_t* d2, int n2, int img2_width, int img2_height, int min_level, int max_level,
                          float radius, int th_low, float nratio, int histo_len, int check_orientation, tb_match* out,
                          int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !count || n1 < 0 || n2 < 0 || histo_len < 1 || img2_width < 1 || img2_height < 1) return TB_EINVAL;
    *count = 0;
    if (n1 == 0) return TB_OK;
    if ((n1 && (!k1 || !d1)) || (n2 && (!k2 || !d2))) return TB_EINVAL;
    const int GRID_ROWS = 36, GRID_COLS = 120;
    /* Frame.cpp:30-31: the two inverse factors are swapped in the reference; kept */
    const float heightInv = (float)GRID_COLS / (float)img2_width;
    const float widthInv = (float)GRID_ROWS / (float)img2_height;
    /* Frame::AssignFeaturesToGrid as CSR (insertion order inside a cell = key index order) */
    std::vector<int32_t> cellOf((size_t)n2), start((size_t)GRID_COLS * GRID_ROWS + 1, 0), items((size_t)std::max(n2, 1));
    for (int i = 0; i < n2; i++) {
        const int posX = (int)roundf(k2[i].x * widthInv), posY = (int)roundf(k2[i].y * heightInv);
        cellOf[i] = (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) ? -1 : posX * GRID_ROWS + posY;
        if (cellOf[i] >= 0) start[cellOf[i] + 1]++;
    }
    for (size_t c = 0; c < (size_t)GRID_COLS * GRID_ROWS; c++) start[c + 1] += start[c];
    {
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < n2; i++)
            if (cellOf[i] >= 0) items[fill[cellOf[i]]++] = i;
    }
    void *dk1, *dd1, *dk2, *dd2, *dst, *dit, *dbest;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)n1 * 32, &dd1))) return rc;
    if ((rc = tb_scratch(ctx, 1, (size_t)std::max(n2, 1) * 32, &dd2))) return rc;
    if ((rc = tb_scratch(ctx, 2, (size_t)n1 * sizeof(tb_keypoint), &dk1))) return rc;
    if ((rc = tb_scratch(ctx, 3, (size_t)std::max(n2, 1) * sizeof(tb_keypoint), &dk2))) return rc;
    if ((rc = tb_scratch(ctx, 4, start.size() * 4, &dst))) return rc;
    if ((rc = tb_scratch(ctx, 5, items.size() * 4, &dit))) return rc;
    if ((rc = tb_scratch(ctx, 6, (size_t)n1 * 16, &dbest))) return rc;
    TB_HIP(ctx, hipMemcpyAsync(dd1, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dk1, k1, (size_t)n1 * sizeof(tb_keypoint), hipMemcpyHostToDevice, ctx->stream));
    if (n2 > 0) {
        TB_HIP(ctx, hipMemcpyAsync(dd2, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, ctx->stream));
        TB_HIP(ctx, hipMemcpyAsync(dk2, k2, (size_t)n2 * sizeof(tb_keypoint), hipMemcpyHostToDevice, ctx->stream));
    }
    TB_HIP(ctx, hipMemcpyAsync(dst, start.data(), start.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dit, items.data(), items.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = tbk_window_match(ctx, (const tb_keypoint*)dk1, (const uint8_t*)dd1, n1, (const tb_keypoint*)dk2, (const uint8_t*)dd2, n2,
                          (const int32_t*)dst, (const int32_t*)dit, widthInv, heightInv, min_level, max_level, radius,
                          (int32_t*)dbest);
    if (rc) return rc;
    std::vector<int32_t> best((size_t)n1 * 4);
    TB_HIP(ctx, hipMemcpyAsync(best.data(), dbest, best.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    /* acceptance + rotation histogram, matcher.cpp:352-392 (bookkeeping over <= n1 survivors) */
    std::vector<tb_match> matches;
    std::vector<std::vector<int>> rotHist((size_t)histo_len);
    const float factor = 1.f / (float)histo_len;
    for (int i1 = 0; i1 < n1; i1++) {
        const int bestDist = best[4 * i1], bestDist2 = best[4 * i1 + 1], bestIdx = best[4 * i1 + 2];
        if (best[4 * i1 + 3] == 0) continue;
        if (bestDist <= th_low && (float)bestDist < (float)bestDist2 * nratio) {
            tb_match m = {i1, bestIdx, -1, (float)bestDist};
            matches.push_back(m);
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx].angle;
                if (rot < 0) rot += 360.f;
                int bin = (int)roundf(rot * factor);
                if (bin == histo_len) bin = 0;
                if (bin < 0 || bin >= histo_len) return tb_fail(ctx, TB_EUNSUPPORTED, "rotation bin %d outside histogram (reference asserts)", bin);
                rotHist[bin].push_back((int)matches.size() - 1);
            }
        }
    }
    std::vector<tb_match> good;
    if (check_orientation) {
        std::vector<int> sizes((size_t)histo_len);
        for (int i = 0; i < histo_len; i++) sizes[i] = (int)rotHist[i].size();
        int ind[3] = {-1, -1, -1};
        tb_three_maxima(sizes.data(), histo_len, &ind[0], &ind[1], &ind[2]);
        for (int i = 0; i < histo_len; i++)
            if (i == ind[0] || i == ind[1] || i == ind[2])
                for (int item : rotHist[i]) good.push_back(matches[item]);
    } else {
        good.swap(matches);
    }
    *count = (int)good.size();
    if ((int)good.size() > cap) return tb_fail(ctx, TB_ECAPACITY, "matches: %d, capacity %d", (int)good.size(), cap);
    if (out) std::copy(good.begin(), good.end(), out);
    return TB_OK;
}

/* ---- Matcher::searchByBow (matcher.cpp:619-721). The two frames' DBoW2 feature vectors are inputs. */
int tb_search_by_bow(tb_ctx* ctx, const tb_keypoint* k1, const uint8_t* d1, int n1, const uint32_t* nodes1, const int32_t* start1,
                     const uint32_t* items1, int nn1, const tb_keypoint* k2, const uint8_t* d2, int n2, const uint8_t* has_mp2,
                     const uint32_t* nodes2, const int32_t* start2, const uint32_t* items2, int nn2, int map_point_only, int th_low,
                     float nratio, int histo_len, int check_orientation, tb_match* out, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !count || n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || histo_len < 1 || cap < 0) return TB_EINVAL;
    *count = 0;
    if ((nn1 && (!nodes1 || !start1)) || (nn2 && (!nodes2 || !start2)) || (n1 && (!k1 || !d1)) || (n2 && (!k2 || !d2))) return TB_EINVAL;
    /* the walk of the two sorted node lists (matcher.cpp:637-698): one query per feature of F1 in a shared node, in the
     * reference's emission order */
    struct Q { int32_t idx1, s2, e2, pad; };
    std::vector<Q> queries;
    int a = 0, b = 0;
    const int tot2 = nn2 ? start2[nn2] : 0;
    while (a < nn1 && b < nn2) {
        if (nodes1[a] == nodes2[b]) {
            if (start2[b] < 0 || start2[b + 1] < start2[b] || start2[b + 1] > tot2 || start1[a + 1] < start1[a]) return tb_fail(ctx, TB_EINVAL, "searchByBow: feature vector offsets");
            for (int p1 = start1[a]; p1 < start1[a + 1]; p1++) {
                const int idx1 = (int)items1[p1];
                if (idx1 < 0 || idx1 >= n1) return tb_fail(ctx, TB_EINVAL, "searchByBow: feature index %d of F1 out of range", idx1);
                queries.push_back(Q{idx1, start2[b], start2[b + 1], 0});
            }
            a++; b++;
        } else if (nodes1[a] < nodes2[b]) {
            while (a < nn1 && nodes1[a] < nodes2[b]) a++;
        } else {
            while (b < nn2 && nodes2[b] < nodes1[a]) b++;
        }
    }
    for (int i = 0; i < tot2; i++)
        if ((int)items2[i] < 0 || (int)items2[i] >= n2) return tb_fail(ctx, TB_EINVAL, "searchByBow: feature index %u of F2 out of range", items2[i]);
    const int nq = (int)queries.size();
    if (nq == 0) return TB_OK;
    void *dd1, *dd2, *dit, *dq, *dbest, *dmp;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)n1 * 32, &dd1)) || (rc = tb_scratch(ctx, 1, (size_t)n2 * 32, &dd2)) ||
        (rc = tb_scratch(ctx, 4, (size_t)std::max(tot2, 1) * 4, &dit)) || (rc = tb_scratch(ctx, 5, (size_t)nq * 16, &dq)) ||
        (rc = tb_scratch(ctx, 6, (size_t)nq * 16, &dbest)) || (rc = tb_scratch(ctx, 3, (size_t)std::max(n2, 1), &dmp)))
        return rc;
    hipStream_t s = ctx->stream;
    TB_HIP(ctx, hipMemcpyAsync(dd1, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(dd2, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(dit, items2, (size_t)tot2 * 4, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(dq, queries.data(), (size_t)nq * 16, hipMemcpyHostToDevice, s));
    if (has_mp2) TB_HIP(ctx, hipMemcpyAsync(dmp, has_mp2, (size_t)n2, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipStreamSynchronize(s)); /* queries is a function-lifetime staging buffer */
    if ((rc = tbk_bow_search(ctx, nq, dq, (const uint8_t*)dd1, (const uint8_t*)dd2, (const uint32_t*)dit, has_mp2 ? (const uint8_t*)dmp : nullptr,
                             map_point_only, dbest)))
        return rc;
    std::vector<int32_t> best((size_t)nq * 4);
    TB_HIP(ctx, hipMemcpyAsync(best.data(), dbest, best.size() * 4, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipStreamSynchronize(s));
    /* acceptance + rotation histogram, matcher.cpp:671-717 (bookkeeping over <= nq survivors) */
    std::vector<tb_match> matches;
    std::vector<std::vector<int>> rotHist((size_t)histo_len);
    const float factor = 1.f / (float)histo_len;
    for (int q = 0; q < nq; q++) {
        const int bestDist1 = best[4 * q], bestDist2 = best[4 * q + 1], bestIdx2 = best[4 * q + 2];
        if (bestDist1 < th_low && bestIdx2 >= 0 && (float)bestDist1 < nratio * (float)bestDist2) {
            const int idx1 = queries[q].idx1;
            tb_match m = {idx1, bestIdx2, -1, (float)bestDist1};
            matches.push_back(m);
            if (check_orientation) {
                float rot = k1[idx1].angle - k2[bestIdx2].angle;
                if (rot < 0) rot += 360.f;
                int bin = (int)roundf(rot * factor);
                if (bin == histo_len) bin = 0;
                if (bin < 0 || bin >= histo_len) return tb_fail(ctx, TB_EUNSUPPORTED, "rotation bin %d outside histogram (reference asserts)", bin);
                rotHist[bin].push_back((int)matches.size() - 1);
            }
        }
    }
    std::vector<tb_match> good;
    if (check_orientation) {
        std::vector<int> sizes((size_t)histo_len);
        for (int i = 0; i < histo_len; i++) sizes[i] = (int)rotHist[i].size();
        int ind[3] = {-1, -1, -1};
        tb_three_maxima(sizes.data(), histo_len, &ind[0], &ind[1], &ind[2]);
        for (int i = 0; i < histo_len; i++)
            if (i == ind[0] || i == ind[1] || i == ind[2])
                for (int item : rotHist[i]) good.push_back(matches[item]);
    } else {
        good.swap(matches);
    }
    *count = (int)good.size();
    if ((int)good.size() > cap) return tb_fail(ctx, TB_ECAPACITY, "matches: %d, capacity %d", (int)good.size(), cap);
    if (out) std::copy(good.begin(), good.end(), out);
    return TB_OK;
}

/* ---- Matcher::searchByProjection, both overloads (matcher.cpp:406-617) */
namespace {
struct ProjHost {
    std::vector<int32_t> best;
    int bad_octave = 0;
};
/* upload F1 (keys, descriptors, taken flags, lookup grid) and the map points, run projection + window search */
int projection_run(tb_ctx* ctx, int map_overload, const float Tcw1[16], const tb_camera* cam1, int img1_w, int img1_h,
                   const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1, const tb_keypoint* k2,
                   const tb_mappoint* mps, const uint8_t* mp_desc, int nq, const float* sf, int nlevels, float nratio,
                   ProjHost& H) {
    const int GRID_ROWS = 36, GRID_COLS = 120;
    const float heightInv = (float)GRID_COLS / (float)img1_w; /* swapped in the reference (Frame.cpp:30-31); kept */
    const float widthInv = (float)GRID_ROWS / (float)img1_h;
    std::vector<int32_t> cellOf((size_t)std::max(n1, 1)), start((size_t)GRID_COLS * GRID_ROWS + 1, 0), items((size_t)std::max(n1, 1));
    for (int i = 0; i < n1; i++) {
        const int posX = (int)roundf(k1[i].x * widthInv), posY = (int)roundf(k1[i].y * heightInv);
        cellOf[i] = (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) ? -1 : posX * GRID_ROWS + posY;
        if (cellOf[i] >= 0) start[cellOf[i] + 1]++;
    }
    for (size_t c = 0; c < (size_t)GRID_COLS * GRID_ROWS; c++) start[c + 1] += start[c];
    {
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < n1; i++)
            if (cellOf[i] >= 0) items[fill[cellOf[i]]++] = i;
    }
    const size_t m1 = (size_t)std::max(n1, 1), mq = (size_t)nq;
    /* slot 7 holds the small arrays back to back (16-byte aligned pieces) */
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t oMp = 0, oTaken = al(oMp + mq * sizeof(tb_mappoint)), oSf = al(oTaken + m1), oQ = al(oSf + (size_t)nlevels * 4),
                 oFlag = al(oQ + mq * 20), misc = oFlag + 16;
    void *dd1, *dmd, *dk1, *dk2, *dst, *dit, *dbest, *dmisc;
    int rc;
    if ((rc = tb_scratch(ctx, 0, m1 * 32, &dd1))) return rc;
    if ((rc = tb_scratch(ctx, 1, mq * 32, &dmd))) return rc;
    if ((rc = tb_scratch(ctx, 2, m1 * sizeof(tb_keypoint), &dk1))) return rc;
    if ((rc = tb_scratch(ctx, 3, mq * sizeof(tb_keypoint), &dk2))) return rc;
    if ((rc = tb_scratch(ctx, 4, start.size() * 4, &dst))) return rc;
    if ((rc = tb_scratch(ctx, 5, items.size() * 4, &dit))) return rc;
    if ((rc = tb_scratch(ctx, 6, mq * 24, &dbest))) return rc;
    if ((rc = tb_scratch(ctx, 7, misc, &dmisc))) return rc;
    char* mb = (char*)dmisc;
    hipStream_t s = ctx->stream;
    std::vector<uint8_t> taken(m1, 0);
    if (taken1) std::copy(taken1, taken1 + n1, taken.begin());
    if (n1 > 0) {
        TB_HIP(ctx, hipMemcpyAsync(dd1, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
        TB_HIP(ctx, hipMemcpyAsync(dk1, k1, (size_t)n1 * sizeof(tb_keypoint), hipMemcpyHostToDevice, s));
    }
    TB_HIP(ctx, hipMemcpyAsync(mb + oTaken, taken.data(), m1, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(dmd, mp_desc, mq * 32, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(mb + oMp, mps, mq * sizeof(tb_mappoint), hipMemcpyHostToDevice, s));
    if (k2) TB_HIP(ctx, hipMemcpyAsync(dk2, k2, mq * sizeof(tb_keypoint), hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(mb + oSf, sf, (size_t)nlevels * 4, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemsetAsync(mb + oFlag, 0, 16, s));
    TB_HIP(ctx, hipMemcpyAsync(dst, start.data(), start.size() * 4, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(dit, items.data(), items.size() * 4, hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipStreamSynchronize(s)); /* the host vectors above go out of use */
    rc = tbk_projection_search(ctx, map_overload, Tcw1, cam1, (const tb_keypoint*)dk2, (const tb_mappoint*)(mb + oMp),
                               (const uint8_t*)dmd, nq, (const float*)(mb + oSf), nlevels, sf[0], nratio, (const tb_keypoint*)dk1,
                               (const uint8_t*)dd1, (const uint8_t*)(mb + oTaken), (const int32_t*)dst, (const int32_t*)dit, widthInv,
                               heightInv, mb + oQ, (int32_t*)dbest, (int*)(mb + oFlag));
    if (rc) return rc;
    H.best.resize(mq * 6);
    TB_HIP(ctx, hipMemcpyAsync(H.best.data(), dbest, mq * 24, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipMemcpyAsync(&H.bad_octave, mb + oFlag, sizeof(int), hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipStreamSynchronize(s));
    return TB_OK;
}
}  // namespace

int tb_search_by_projection(tb_ctx* ctx, const float Tcw1[16], const tb_camera* cam1, int img1_width, int img1_height,
                            const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1, const tb_keypoint* k2,
                            const tb_mappoint* mp2, const uint8_t* mp2_desc, int n2, const float* scale_factors, int nlevels,
                            float nratio, int th_high, int histo_len, int check_orientation, tb_match* out, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !count || !Tcw1 || !cam1 || n1 < 0 || n2 < 0 || histo_len < 1 || nlevels < 1 || !scale_factors || img1_width < 1 ||
        img1_height < 1)
        return TB_EINVAL;
    *count = 0;
    if (n2 == 0) return TB_OK;
    if ((n1 && (!k1 || !d1)) || !k2 || !mp2 || !mp2_desc) return TB_EINVAL;
    ProjHost H;
    int rc = projection_run(ctx, 0, Tcw1, cam1, img1_width, img1_height, k1, d1, taken1, n1, k2, mp2, mp2_desc, n2, scale_factors,
                            nlevels, nratio, H);
    if (rc) return rc;
    if (H.bad_octave) return tb_fail(ctx, TB_EINVAL, "searchByProjection: a key octave is outside the %d scale factors", nlevels);
    /* acceptance + rotation histogram, matcher.cpp:483-530 (bookkeeping over <= n2 survivors) */
    std::vector<tb_match> matches;
    std::vector<std::vector<int>> rotHist((size_t)histo_len);
    const float factor = 1.0f / (float)histo_len;
    for (int i2 = 0; i2 < n2; i2++) {
        const int bestDist = H.best[6 * (size_t)i2], bestIdx1 = H.best[6 * (size_t)i2 + 2];
        if (H.best[6 * (size_t)i2 + 5] == 0 || bestIdx1 < 0) continue;
        if (bestDist <= th_high) {
            tb_match m = {bestIdx1, i2, -1, (float)bestDist};
            matches.push_back(m);
            if (check_orientation) {
                float rot = k2[i2].angle - k1[bestIdx1].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == histo_len) bin = 0;
                if (bin < 0 || bin >= histo_len) return tb_fail(ctx, TB_EUNSUPPORTED, "rotation bin %d outside histogram (reference asserts)", bin);
                rotHist[bin].push_back((int)matches.size() - 1);
            }
        }
    }
    std::vector<tb_match> good;
    if (check_orientation) {
        std::vector<int> sizes((size_t)histo_len);
        for (int i = 0; i < histo_len; i++) sizes[i] = (int)rotHist[i].size();
        int ind[3] = {-1, -1, -1};
        tb_three_maxima(sizes.data(), histo_len, &ind[0], &ind[1], &ind[2]);
        for (int i = 0; i < histo_len; i++)
            if (i == ind[0] || i == ind[1] || i == ind[2])
                for (int item : rotHist[i]) good.push_back(matches[item]);
    } else {
        good.swap(matches);
    }
    *count = (int)good.size();
    if ((int)good.size() > cap) return tb_fail(ctx, TB_ECAPACITY, "matches: %d, capacity %d", (int)good.size(), cap);
    if (out) std::copy(good.begin(), good.end(), out);
    return TB_OK;
}

int tb_search_by_projection_map(tb_ctx* ctx, const float Tcw1[16], const tb_camera* cam1, int img1_width, int img1_height,
                                const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1, int n1, const tb_mappoint* mps,
                                const uint8_t* mp_desc, int nmp, const float* scale_factors, int nlevels, float nratio, float radio,
                                int th_high, tb_match* out, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !count || !Tcw1 || !cam1 || n1 < 0 || nmp < 0 || nlevels < 1 || !scale_factors || img1_width < 1 || img1_height < 1)
        return TB_EINVAL;
    *count = 0;
    if (nmp == 0) return TB_OK;
    if ((n1 && (!k1 || !d1)) || !mps || !mp_desc) return TB_EINVAL;
    ProjHost H;
    int rc = projection_run(ctx, 1, Tcw1, cam1, img1_width, img1_height, k1, d1, taken1, n1, nullptr, mps, mp_desc, nmp,
                            scale_factors, nlevels, nratio, H);
    if (rc) return rc;
    std::vector<tb_match> matches;
    for (int im = 0; im < nmp; im++) { /* ratio test, matcher.cpp:606-613 */
        const int32_t* b = &H.best[6 * (size_t)im];
        if (b[5] == 0 || b[2] < 0) continue;
        if (b[0] <= th_high) {
            if (b[3] == b[4] && (float)b[0] > radio * (float)b[1]) continue;
            tb_match m = {b[2], im, -1, (float)b[0]};
            matches.push_back(m);
        }
    }
    *count = (int)matches.size();
    if ((int)matches.size() > cap) return tb_fail(ctx, TB_ECAPACITY, "matches: %d, capacity %d", (int)matches.size(), cap);
    if (out) std::copy(matches.begin(), matches.end(), out);
    return TB_OK;
}

/* ---- device-resident lookup grid + batched projection search */
int tb_frame_grid_batch_dev(tb_ctx* ctx, int nframes, const tb_keypoint* keys, const int32_t* counts, int key_pitch, int img_width,
                            int img_height, int32_t* cell_start, int32_t* cell_items) {
    TB_ENTER(ctx);
    if (!ctx || nframes < 0 || key_pitch < 1 || img_width < 1 || img_height < 1 || (nframes && (!keys || !counts || !cell_start || !cell_items)))
        return TB_EINVAL;
    return tbk_grid_build_batch(ctx, nframes, keys, counts, key_pitch, img_width, img_height, cell_start, cell_items);
}

int tb_search_by_projection_batch_dev(tb_ctx* ctx, int npairs, const float* Tcw1, const tb_camera* cam1, int img1_width,
                                      int img1_height, const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1,
                                      const int32_t* n1, int pitch1, const int32_t* cell_start, const int32_t* cell_items,
                                      const tb_keypoint* k2, const tb_mappoint* mp2, const uint8_t* mp2_desc, const int32_t* n2,
                                      int pitch2, const float* scale_factors, int nlevels, float nratio, int th_high, int histo_len,
                                      int check_orientation, tb_match* out, int cap, int32_t* out_counts, int32_t* flags) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || !cam1 || !scale_factors || nlevels < 1 || nlevels > TB_MAX_LEVELS * 2 || histo_len < 1 || histo_len > 1024 ||
        pitch1 < 1 || pitch2 < 1 || cap < 0 || img1_width < 1 || img1_height < 1)
        return TB_EINVAL;
    if (npairs == 0) return TB_OK;
    if (!Tcw1 || !k1 || !d1 || !taken1 || !n1 || !cell_start || !cell_items || !k2 || !mp2 || !mp2_desc || !n2 || !out || !out_counts || !flags)
        return TB_EINVAL;
    void* dbest;
    int rc = tb_scratch(ctx, 6, (size_t)npairs * pitch2 * 6 * sizeof(int32_t), &dbest);
    if (rc) return rc;
    return tbk_projection_batch(ctx, npairs, Tcw1, cam1, img1_width, img1_height, k1, d1, taken1, n1, pitch1, cell_start, cell_items, k2, mp2,
                                mp2_desc, n2, pitch2, scale_factors, nlevels, nratio, th_high, histo_len, check_orientation,
                                (int32_t*)dbest, out, cap, out_counts, flags, 0, 0.f, pitch2);
}

int tb_search_by_projection_map_batch_dev(tb_ctx* ctx, int npairs, const float* Tcw1, const tb_camera* cam1, int img1_width,
                                          int img1_height, const tb_keypoint* k1, const uint8_t* d1, const uint8_t* taken1,
                                          const int32_t* n1, int pitch1, const int32_t* cell_start, const int32_t* cell_items,
                                          const tb_mappoint* mps, const uint8_t* mp_desc, const int32_t* nmp, int mp_pitch,
                                          int max_nmp, const float* scale_factors, int nlevels, float nratio, float radio,
                                          int th_high, tb_match* out, int cap, int32_t* out_counts, int32_t* flags) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || !cam1 || !scale_factors || nlevels < 1 || nlevels > TB_MAX_LEVELS * 2 || pitch1 < 1 || mp_pitch < 0 ||
        max_nmp < 1 || (mp_pitch > 0 && mp_pitch < max_nmp) || cap < 0 || img1_width < 1 || img1_height < 1)
        return TB_EINVAL;
    if (npairs == 0) return TB_OK;
    if (!Tcw1 || !k1 || !d1 || !taken1 || !n1 || !cell_start || !cell_items || !mps || !mp_desc || !nmp || !out || !out_counts || !flags)
        return TB_EINVAL;
    void* dbest;
    int rc = tb_scratch(ctx, 6, (size_t)npairs * max_nmp * 6 * sizeof(int32_t), &dbest);
    if (rc) return rc;
    return tbk_projection_batch(ctx, npairs, Tcw1, cam1, img1_width, img1_height, k1, d1, taken1, n1, pitch1, cell_start, cell_items,
                                nullptr, mps, mp_desc, nmp, mp_pitch, scale_factors, nlevels, nratio, th_high, 1, 0, (int32_t*)dbest, out,
                                cap, out_counts, flags, 1, radio, max_nmp);
}

int tb_search_by_violence_batch_dev(tb_ctx* ctx, int npairs, const tb_keypoint* k1, const uint8_t* d1, const int32_t* n1, int pitch1,
                                    const tb_keypoint* k2, const uint8_t* d2, const int32_t* n2, int pitch2,
                                    const int32_t* cell_start2, const int32_t* cell_items2, int img2_width, int img2_height,
                                    int min_level, int max_level, float radius, int th_low, float nratio, int histo_len,
                                    int check_orientation, tb_match* out, int cap, int32_t* out_counts, int32_t* flags) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || histo_len < 1 || histo_len > 1024 || pitch1 < 1 || pitch2 < 1 || cap < 0 || img2_width < 1 || img2_height < 1)
        return TB_EINVAL;
    if (npairs == 0) return TB_OK;
    if (!k1 || !d1 || !n1 || !k2 || !d2 || !n2 || !cell_start2 || !cell_items2 || !out || !out_counts || !flags) return TB_EINVAL;
    void* dbest;
    int rc = tb_scratch(ctx, 6, (size_t)npairs * pitch1 * 4 * sizeof(int32_t), &dbest);
    if (rc) return rc;
    return tbk_violence_batch(ctx, npairs, k1, d1, n1, pitch1, k2, d2, n2, pitch2, cell_start2, cell_items2, img2_width, img2_height,
                              min_level, max_level, radius, th_low, nratio, histo_len, check_orientation, (int32_t*)dbest, out, cap,
                              out_counts, flags);
}

/* ------------------------------------------------------------------ pose optimisation / local BA */
int tb_pose_opt_batch_dev(tb_ctx* ctx, int nproblems, const double K[4], const float* Tcw_in, const tb_obs* obs,
                          const int32_t* counts, int obs_pitch, uint8_t* outlier, float* Tcw_out, int32_t* n_inliers,
                          double* stats) {
    TB_ENTER(ctx);
    if (!ctx || nproblems < 0 || !K || !Tcw_in || !obs || !counts || !outlier || !Tcw_out || !n_inliers || obs_pitch < 1)
        return TB_EINVAL;
    void* derr;
    int rc = tb_scratch(ctx, 7, (size_t)nproblems * obs_pitch * 3 * sizeof(double), &derr);
    if (rc) return rc;
    return tbk_pose_batch(ctx, nproblems, K, Tcw_in, obs, counts, obs_pitch, outlier, Tcw_out, n_inliers, stats, (double*)derr);
}

int tb_pose_opt(tb_ctx* ctx, const double K[4], const float Tcw_in[16], const tb_obs* obs, int n, uint8_t* outlier,
                float Tcw_out[16], int* n_inliers, double* stats) {
    TB_ENTER(ctx);
    if (!ctx || !K || !Tcw_in || !Tcw_out || !n_inliers || n < 0 || (n && (!obs || !outlier))) return TB_EINVAL;
    const int pitch = std::max(n, 1);
    void *dobs, *dmisc, *dout;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)pitch * sizeof(tb_obs), &dobs))) return rc;
    if ((rc = tb_scratch(ctx, 1, (size_t)pitch, &dout))) return rc;
    if ((rc = tb_scratch(ctx, 2, 512, &dmisc))) return rc;
    /* dmisc: Tin[16] f32 | Tout[16] f32 | count i32 | ninl i32 | stats[8] f64 (at byte 192) */
    float* dTin = (float*)dmisc;
    float* dTout = dTin + 16;
    int32_t* dcnt = (int32_t*)(dTout + 16);
    int32_t* dninl = dcnt + 1;
    double* dstats = (double*)((char*)dmisc + 192);
    int32_t cnt = n;
    if (n > 0) {
        TB_HIP(ctx, hipMemcpyAsync(dobs, obs, (size_t)n * sizeof(tb_obs), hipMemcpyHostToDevice, ctx->stream));
        TB_HIP(ctx, hipMemcpyAsync(dout, outlier, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    TB_HIP(ctx, hipMemcpyAsync(dTin, Tcw_in, 64, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dcnt, &cnt, 4, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = tb_pose_opt_batch_dev(ctx, 1, K, dTin, (const tb_obs*)dobs, dcnt, pitch, (uint8_t*)dout, dTout, dninl, dstats);
    if (rc) return rc;
    int32_t ninl = 0;
    TB_HIP(ctx, hipMemcpyAsync(Tcw_out, dTout, 64, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(&ninl, dninl, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (n > 0) TB_HIP(ctx, hipMemcpyAsync(outlier, dout, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (stats) TB_HIP(ctx, hipMemcpyAsync(stats, dstats, 64, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_inliers = ninl;
    return TB_OK;
}

int tb_local_ba_batch_dev(tb_ctx* ctx, int nwindows, const double K[4], int nkf, int nfixed, float* poses, int npt, float* pts,
                          const tb_ba_obs* obs, const int32_t* obs_counts, int obs_pitch, int iters, double* stats) {
    TB_ENTER(ctx);
    if (!ctx || !K || !poses || !pts || !obs || !obs_counts || nwindows < 0 || nkf < 1 || npt < 1 || obs_pitch < 1 || nfixed < 0 ||
        nfixed > nkf || iters < 0)
        return TB_EINVAL;
    if (nwindows == 0) return TB_OK;
    const size_t wb = tbk_local_ba_work_bytes(ctx, nwindows, nkf, nfixed, npt, obs_pitch);
    void* dwork;
    int rc = tb_scratch(ctx, 6, wb, &dwork);
    if (rc) return rc;
    return tbk_local_ba_batch(ctx, nwindows, K, nkf, nfixed, poses, npt, pts, obs, obs_counts, obs_pitch, iters, stats, dwork, wb);
}

int tb_local_ba(tb_ctx* ctx, const double K[4], int nkf, int nfixed, float* poses, int npt, float* pts, const tb_ba_obs* obs,
                int nobs, int iters, double* stats) {
    TB_ENTER(ctx);
    if (!ctx || !K || !poses || !pts || !obs || nkf < 1 || npt < 1 || nobs < 1 || nfixed < 0 || nfixed > nkf || iters < 0)
        return TB_EINVAL;
    for (int e = 0; e < nobs; e++)
        if (obs[e].kf < 0 || obs[e].kf >= nkf || obs[e].pt < 0 || obs[e].pt >= npt)
            return tb_fail(ctx, TB_EINVAL, "local_ba: observation %d out of range", e);
    /* the kernels want observations grouped by point: stable sort keeps each point's edges in caller order */
    std::vector<tb_ba_obs> sorted(obs, obs + nobs);
    std::stable_sort(sorted.begin(), sorted.end(), [](const tb_ba_obs& a, const tb_ba_obs& b) { return a.pt < b.pt; });
    void *dposes, *dpts, *dobs, *dmisc;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)nkf * 64, &dposes))) return rc;
    if ((rc = tb_scratch(ctx, 1, (size_t)npt * 12, &dpts))) return rc;
    if ((rc = tb_scratch(ctx, 2, (size_t)nobs * sizeof(tb_ba_obs), &dobs))) return rc;
    if ((rc = tb_scratch(ctx, 3, 256, &dmisc))) return rc;
    double* dstats = (double*)dmisc;
    int32_t* dcnt = (int32_t*)((char*)dmisc + 64);
    int32_t cnt = nobs;
    TB_HIP(ctx, hipMemcpyAsync(dposes, poses, (size_t)nkf * 64, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dpts, pts, (size_t)npt * 12, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dobs, sorted.data(), (size_t)nobs * sizeof(tb_ba_obs), hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dcnt, &cnt, 4, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    rc = tb_local_ba_batch_dev(ctx, 1, K, nkf, nfixed, (float*)dposes, npt, (float*)dpts, (const tb_ba_obs*)dobs, dcnt, nobs, iters, dstats);
    if (rc) return rc;
    double st[8];
    TB_HIP(ctx, hipMemcpyAsync(poses, dposes, (size_t)nkf * 64, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(pts, dpts, (size_t)npt * 12, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(st, dstats, 64, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) memcpy(stats, st, 64);
    if (st[7] < 0) return tb_fail(ctx, TB_EINVAL, "local_ba: observations rejected by the device-side check");
    return TB_OK;
}

int tb_clahe_dev(tb_ctx* ctx, const uint8_t* src, int width, int height, int stride, double clip_limit, int tiles_x, int tiles_y,
                 uint8_t* dst, int dst_stride) {
    TB_ENTER(ctx);
    if (!ctx || !src || !dst || width < 1 || height < 1 || stride < width || dst_stride < width || tiles_x < 1 || tiles_y < 1) return TB_EINVAL;
    void* lut;
    int rc;
    if ((rc = tb_scratch(ctx, 6, (size_t)tiles_x * tiles_y * 256, &lut))) return rc;
    return tbk_clahe(ctx, 1, src, width, height, stride, 0, clip_limit, tiles_x, tiles_y, dst, dst_stride, 0, (uint8_t*)lut);
}

int tb_clahe(tb_ctx* ctx, const uint8_t* src, int width, int height, int stride, double clip_limit, int tiles_x, int tiles_y,
             uint8_t* dst, int dst_stride) {
    TB_ENTER(ctx);
    if (!ctx || !src || !dst || width < 1 || height < 1 || stride < width || dst_stride < width || tiles_x < 1 || tiles_y < 1) return TB_EINVAL;
    void *ds, *dd;
    int rc;
    if ((rc = tb_scratch(ctx, 0, (size_t)stride * height, &ds))) return rc;
    if ((rc = tb_scratch(ctx, 1, (size_t)dst_stride * height, &dd))) return rc;
    TB_HIP(ctx, hipMemcpyAsync(ds, src, (size_t)stride * height, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = tb_clahe_dev(ctx, (const uint8_t*)ds, width, height, stride, clip_limit, tiles_x, tiles_y, (uint8_t*)dd, dst_stride))) return rc;
    TB_HIP(ctx, hipMemcpy2DAsync(dst, dst_stride, dd, dst_stride, width, height, hipMemcpyDeviceToHost, ctx->stream));
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TB_OK;
}

int tb_optical_flow_pyr_lk_dev(tb_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height, int stride,
                               const float* prev_pts, int n, int win, int max_level, float* next_pts, uint8_t* status, float* err) {
    TB_ENTER(ctx);
    if (!ctx || !prev || !next || n < 0 || width < 1 || height < 1 || stride < width) return TB_EINVAL;
    if (n && (!prev_pts || !next_pts || !status)) return TB_EINVAL;
    if (max_level < 0 || max_level > 5) return tb_fail(ctx, TB_EUNSUPPORTED, "optical flow: max_level %d (0..5)", max_level);
    void* work;
    int rc;
    if ((rc = tb_scratch(ctx, 7, tbk_lk_work_bytes(width, height, max_level, 1), &work))) return rc;
    return tbk_lk_track(ctx, 1, prev, next, width, height, stride, 0, prev_pts, nullptr, n, n, win, max_level, next_pts, status, err, work,
                        nullptr);
}

int tb_optical_flow_pyr_lk_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* prev, const uint8_t* next, int width, int height,
                                     int stride, size_t image_pitch, const float* prev_pts, const int32_t* counts, int pts_pitch,
                                     int win, int max_level, float* next_pts, uint8_t* status, float* err) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || pts_pitch < 0 || width < 1 || height < 1 || stride < width) return TB_EINVAL;
    if (npairs == 0 || pts_pitch == 0) return TB_OK;
    if (!prev || !next || !prev_pts || !next_pts || !status || image_pitch < (size_t)stride * height) return TB_EINVAL;
    if (max_level < 0 || max_level > 5) return tb_fail(ctx, TB_EUNSUPPORTED, "optical flow: max_level %d (0..5)", max_level);
    void* work;
    int rc;
    if ((rc = tb_scratch(ctx, 7, tbk_lk_work_bytes(width, height, max_level, npairs), &work))) return rc;
    return tbk_lk_track(ctx, npairs, prev, next, width, height, stride, image_pitch, prev_pts, counts, pts_pitch, pts_pitch, win, max_level,
                        next_pts, status, err, work, nullptr);
}

int tb_optical_flow_pyr_lk(tb_ctx* ctx, const uint8_t* prev, const uint8_t* next, int width, int height, int stride,
                           const float* prev_pts, int n, int win, int max_level, float* next_pts, uint8_t* status, float* err,
                           int* top_level) {
    TB_ENTER(ctx);
    if (!ctx || !prev || !next || n < 0 || width < 1 || height < 1 || stride < width) return TB_EINVAL;
    if (n && (!prev_pts || !next_pts || !status)) return TB_EINVAL;
    if (max_level < 0 || max_level > 5) return tb_fail(ctx, TB_EUNSUPPORTED, "optical flow: max_level %d (0..5)", max_level);
    const size_t img = (size_t)stride * height, np2 = (size_t)std::max(n, 1) * 2 * sizeof(float);
    void *dp, *dn, *dpts, *dout, *dst, *derr, *work;
    int rc;
    if ((rc = tb_scratch(ctx, 0, img, &dp))) return rc;
    if ((rc = tb_scratch(ctx, 1, img, &dn))) return rc;
    if ((rc = tb_scratch(ctx, 2, np2, &dpts))) return rc;
    if ((rc = tb_scratch(ctx, 3, np2, &dout))) return rc;
    if ((rc = tb_scratch(ctx, 4, (size_t)std::max(n, 1), &dst))) return rc;
    if ((rc = tb_scratch(ctx, 5, (size_t)std::max(n, 1) * sizeof(float), &derr))) return rc;
    if ((rc = tb_scratch(ctx, 7, tbk_lk_work_bytes(width, height, max_level, 1), &work))) return rc;
    TB_HIP(ctx, hipMemcpyAsync(dp, prev, img, hipMemcpyHostToDevice, ctx->stream));
    TB_HIP(ctx, hipMemcpyAsync(dn, next, img, hipMemcpyHostToDevice, ctx->stream));
    if (n) TB_HIP(ctx, hipMemcpyAsync(dpts, prev_pts, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int top = 0;
    rc = tbk_lk_track(ctx, 1, (const uint8_t*)dp, (const uint8_t*)dn, width, height, stride, 0, (const float*)dpts, nullptr, n, n, win,
                      max_level, (float*)dout, (uint8_t*)dst, (float*)derr, work, &top);
    if (rc) return rc;
    if (n) {
        TB_HIP(ctx, hipMemcpyAsync(next_pts, dout, (size_t)n * 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        TB_HIP(ctx, hipMemcpyAsync(status, dst, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        if (err) TB_HIP(ctx, hipMemcpyAsync(err, derr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    TB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (top_level) *top_level = top;
    return TB_OK;
}

int tb_search_by_opflow(tb_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int width, int height, int stride,
                        const tb_camera* cam1, const float* keys2_xy, int n, int equalized, int reject, float* cur_points,
                        tb_match* out, int cap, int* count) {
    TB_ENTER(ctx);
    if (!ctx || !count || !cam1 || n < 0 || cap < 0 || (n && (!cur_points || !keys2_xy)) || (cap && !out)) return TB_EINVAL;
    *count = 0;
    std::vector<uint8_t> status((size_t)std::max(n, 1)), eq;
    int rc;
    if (equalized) { /* matcher.cpp:736-739: img1 = F1->Equalize() = CLAHE(3.0, 8 x 8) of F1's level 0 (Frame.cpp:453-458) */
        eq.resize((size_t)stride * height);
        if ((rc = tb_clahe(ctx, img1, width, height, stride, 3.0, 8, 8, eq.data(), stride))) return rc;
        img1 = eq.data();
    }
    /* matcher.cpp:744: calcOpticalFlowPyrLK(img2, img1, keys of F2, cur_points, ..., Size(21, 21), 3) */
    rc = tb_optical_flow_pyr_lk(ctx, img2, img1, width, height, stride, keys2_xy, n, 21, 3, cur_points, status.data(), nullptr, nullptr);
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        if (!status[i]) continue;
        /* :746-748, IsInFrame(Vector2i(cur.x, cur.y)): the conversion truncates; out-of-int-range converts to INT_MIN on
         * x86 and fails the test */
        const float x = cur_points[2 * i], y = cur_points[2 * i + 1];
        bool in = fabsf(x) < 2147483648.f && fabsf(y) < 2147483648.f;
        if (in) {
            const int u = (int)x, v = (int)y;
            in = u >= 0 && u < (int)((float)cam1->width * 1.f) && v >= 0 && v < (int)((float)cam1->height * 1.f);
        }
        if (!in) status[i] = 0;
    }
    if (reject && (rc = tb_reject_with_f(ctx, cur_points, keys2_xy, n, status.data()))) return rc; /* matcher.cpp:751-755 */
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (!status[i]) continue;
        if (m >= cap) return tb_fail(ctx, TB_ECAPACITY, "searchByOPFlow: more than %d matches", cap);
        out[m].queryIdx = i; out[m].trainIdx = i; out[m].imgIdx = -1; out[m].distance = 3.402823466e+38f; /* cv::DMatch() */
        m++;
    }
    *count = m;
    return TB_OK;
}

int tb_search_by_opflow_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* img1, const uint8_t* img2, int width, int height, int stride,
                                  size_t image_pitch, const tb_camera* cam1, const float* keys2_xy, const int32_t* counts, int pts_pitch,
                                  int equalized, int reject, float* cur_points, uint8_t* status, tb_match* out, int cap,
                                  int32_t* out_counts) {
    TB_ENTER(ctx);
    if (!ctx || !cam1 || npairs < 0 || pts_pitch < 0 || cap < 0 || width < 1 || height < 1 || stride < width) return TB_EINVAL;
    if (npairs == 0) return TB_OK;
    if (!img1 || !img2 || !out_counts || image_pitch < (size_t)stride * height) return TB_EINVAL;
    if (pts_pitch && (!keys2_xy || !cur_points || !status || (cap && !out))) return TB_EINVAL;
    int rc;
    const uint8_t* next = img1;
    if (equalized) { /* matcher.cpp:736-739: img1 = F1->Equalize() (Frame.cpp:453-458) */
        void *eq, *lut;
        if ((rc = tb_scratch(ctx, 5, (size_t)npairs * image_pitch, &eq))) return rc;
        if ((rc = tb_scratch(ctx, 6, (size_t)npairs * 8 * 8 * 256, &lut))) return rc;
        if ((rc = tbk_clahe(ctx, npairs, img1, width, height, stride, image_pitch, 3.0, 8, 8, (uint8_t*)eq, stride, image_pitch, (uint8_t*)lut))) return rc;
        next = (const uint8_t*)eq;
    }
    if (pts_pitch) {
        void* work;
        if ((rc = tb_scratch(ctx, 7, tbk_lk_work_bytes(width, height, 3, npairs), &work))) return rc;
        /* matcher.cpp:744: calcOpticalFlowPyrLK(img2, img1, keys of F2, cur_points, ..., Size(21, 21), 3) */
        if ((rc = tbk_lk_track(ctx, npairs, img2, next, width, height, stride, image_pitch, keys2_xy, counts, pts_pitch, pts_pitch, 21, 3,
                               cur_points, status, nullptr, work, nullptr)))
            return rc;
    }
    if ((rc = tbk_flow_accept(ctx, npairs, cur_points, status, counts, pts_pitch, cam1->width, cam1->height, out, cap, out_counts))) return rc;
    if (reject && pts_pitch) {
        /* matcher.cpp:751-755: rejectWithF(cur_points, F2->GetCVKeys(), status); then the matches of what is left (pairs with
         * 8..14 tracked points take cv::findFundamentalMat's LMedS branch inside the same kernel, as in tb_reject_with_f) */
        void *work, *fl;
        if ((rc = tb_scratch(ctx, 8, tbk_ransac_work_bytes(npairs, pts_pitch), &work))) return rc;
        if ((rc = tb_scratch(ctx, 9, (size_t)npairs * sizeof(int32_t), &fl))) return rc;
        if ((rc = tbk_ransac_f(ctx, npairs, cur_points, keys2_xy, status, counts, pts_pitch, 0, 1.0, 0.99, work, (int32_t*)fl, nullptr, nullptr)))
            return rc;
        rc = tbk_flow_accept(ctx, npairs, cur_points, status, counts, pts_pitch, cam1->width, cam1->height, out, cap, out_counts);
    }
    return rc;
}

/* Matcher::rejectWithF / cv::findFundamentalMat, host forms: upload, one workgroup, download */
static int ransac_host(tb_ctx* ctx, const float* p1, const float* p2, int n, uint8_t* status, int mode, double thresh, double conf,
                       double* F, int* iters, int* flag) {
    void *d1, *d2, *dst, *work, *misc;
    int rc;
    const size_t nb = (size_t)std::max(n, 1) * 2 * sizeof(float);
    if ((rc = tb_scratch(ctx, 0, nb, &d1)) || (rc = tb_scratch(ctx, 1, nb, &d2)) || (rc = tb_scratch(ctx, 2, (size_t)std::max(n, 1), &dst)) ||
        (rc = tb_scratch(ctx, 8, tbk_ransac_work_bytes(1, n), &work)) || (rc = tb_scratch(ctx, 9, 256, &misc)))
        return rc;
    hipStream_t s = ctx->stream;
    TB_HIP(ctx, hipMemcpyAsync(d1, p1, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(d2, p2, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, s));
    TB_HIP(ctx, hipMemcpyAsync(dst, status, (size_t)n, hipMemcpyHostToDevice, s));
    int32_t* d_flag = (int32_t*)misc;
    int32_t* d_iters = d_flag + 1;
    double* d_F = (double*)((char*)misc + 16);
    TB_HIP(ctx, hipMemsetAsync(misc, 0, 128, s));
    if ((rc = tbk_ransac_f(ctx, 1, (const float*)d1, (const float*)d2, (uint8_t*)dst, nullptr, n, mode, thresh, conf, work, d_flag, d_F, d_iters)))
        return rc;
    int32_t h[2] = {0, 0};
    double hF[9];
    TB_HIP(ctx, hipMemcpyAsync(status, dst, (size_t)n, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipMemcpyAsync(h, misc, sizeof h, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipMemcpyAsync(hF, d_F, sizeof hF, hipMemcpyDeviceToHost, s));
    TB_HIP(ctx, hipStreamSynchronize(s));
    if (flag) *flag = h[0];
    if (iters) *iters = h[1];
    if (F) memcpy(F, hF, sizeof hF);
    return TB_OK;
}

int tb_find_fundamental_ransac(tb_ctx* ctx, const float* pts1, const float* pts2, int n, double thresh, double conf, uint8_t* mask,
                               double* F, int* iters, int* ok) {
    TB_ENTER(ctx);
    if (!ctx || n < 0 || !ok || (n && (!pts1 || !pts2 || !mask))) return TB_EINVAL;
    *ok = 0;
    if (iters) *iters = 0;
    if (n < 7) return TB_OK;                     /* cv::findFundamentalMat returns an empty matrix and no mask */
    int flag = 0;
    memset(mask, 0, (size_t)n);
    const int rc = ransac_host(ctx, pts1, pts2, n, mask, 1, thresh, conf, F, iters, &flag);
    if (rc) return rc;
    *ok = flag == 0 ? 1 : 0;
    return TB_OK;
}

int tb_reject_with_f_batch_dev(tb_ctx* ctx, int npairs, const float* cur_pts, const float* last_pts, const int32_t* counts,
                               int pts_pitch, uint8_t* status) {
    TB_ENTER(ctx);
    if (!ctx || npairs < 0 || pts_pitch < 0) return TB_EINVAL;
    if (npairs == 0 || pts_pitch == 0) return TB_OK;
    if (!cur_pts || !last_pts || !status) return TB_EINVAL;
    void *work, *fl;
    int rc;
    if ((rc = tb_scratch(ctx, 8, tbk_ransac_work_bytes(npairs, pts_pitch), &work))) return rc;
    if ((rc = tb_scratch(ctx, 9, (size_t)npairs * sizeof(int32_t), &fl))) return rc;
    return tbk_ransac_f(ctx, npairs, cur_pts, last_pts, status, counts, pts_pitch, 0, 1.0, 0.99, work, (int32_t*)fl, nullptr, nullptr);
}

int tb_reject_with_f(tb_ctx* ctx, const float* cur_pts, const float* last_pts, int n, uint8_t* status) {
    TB_ENTER(ctx);
    if (!ctx || n < 0 || (n && (!cur_pts || !last_pts || !status))) return TB_EINVAL;
    if (!(n > 8)) return TB_OK;                  /* matcher.cpp:870: findFundamentalMat is not called */
    int flag = 0;
    const int rc = ransac_host(ctx, cur_pts, last_pts, n, status, 0, 1.0, 0.99, nullptr, nullptr, &flag);
    if (rc) return rc;
    return TB_OK;
}

int tb_add_map_points_by_stereo_batch_dev(tb_ctx* ctx, int npairs, const uint8_t* img_stereo, const uint8_t* img_current, int width,
                                          int height, int stride, size_t image_pitch, const tb_camera* cam_stereo, const float* keys_xy,
                                          const int32_t* counts, int pts_pitch, float bf, float* cur_points, uint8_t* status,
                                          float* depth) {
    TB_ENTER(ctx);
    if (!ctx || !cam_stereo || npairs < 0 || pts_pitch < 0) return TB_EINVAL;
    if (npairs == 0 || pts_pitch == 0) return TB_OK;
    if (!depth || !cur_points || !status || !keys_xy) return TB_EINVAL;
    void *m, *mc;
    int rc;
    if ((rc = tb_scratch(ctx, 10, (size_t)npairs * pts_pitch * sizeof(tb_match), &m))) return rc;
    if ((rc = tb_scratch(ctx, 11, (size_t)npairs * sizeof(int32_t), &mc))) return rc;
    /* LocalBA.cpp:54: matcher->searchByOPFlow(stereo_frame, current_frame, pts, true, true) */
    if ((rc = tb_search_by_opflow_batch_dev(ctx, npairs, img_stereo, img_current, width, height, stride, image_pitch, cam_stereo, keys_xy,
                                            counts, pts_pitch, 1, 1, cur_points, status, (tb_match*)m, pts_pitch, (int32_t*)mc)))
        return rc;
    return tbk_stereo_depth(ctx, npairs, cur_points, keys_xy, status, counts, pts_pitch, bf, depth);
}

int tb_add_map_points_by_stereo(tb_ctx* ctx, const uint8_t* img_stereo, const uint8_t* img_current, int width, int height, int stride,
                                const tb_camera* cam_stereo, const float* keys_xy, int n, float bf, float* depth, int* n_depth) {
    TB_ENTER(ctx);
    if (!ctx || !cam_stereo || n < 0 || !n_depth || (n && (!keys_xy || !depth)) || !img_stereo || !img_current) return TB_EINVAL;
    *n_depth = 0;
    for (int i = 0; i < n; i++) depth[i] = -1.0f;
    if (n == 0) return TB_OK;
    std::vector<float> cur((size_t)2 * n);
    std::vector<tb_match> mt((size_t)n);
    int cnt = 0;
    const int rc = tb_search_by_opflow(ctx, img_stereo, img_current, width, height, stride, cam_stereo, keys_xy, n, 1, 1, cur.data(),
                                       mt.data(), n, &cnt);
    if (rc) return rc;
    for (int k = 0; k < cnt; k++) {  /* LocalBA.cpp:58-65: left_id = trainIdx, right_id = queryIdx (equal) */
        const int i = mt[k].trainIdx;
        depth[i] = bf / fabsf(cur[2 * mt[k].queryIdx] - keys_xy[2 * i]);
    }
    *n_depth = cnt;
    return TB_OK;
}

/* ---- multi-GPU batch entry: see include/tb_capi.h */
int tb_batch_run(tb_ctx** ctxs, int ngpu, const tb_batch_params* p, int nframes, const uint8_t* left, const uint8_t* right,
                 int stride, size_t pitch, int cap, tb_keypoint* kps, uint8_t* desc, int32_t* counts, tb_match* matches,
                 int32_t* match_counts) {
    if (!ctxs || ngpu < 1 || !p || nframes < 0 || !kps || !desc || !counts || !matches || !match_counts || cap < 1) return TB_EINVAL;
    for (int i = 0; i < ngpu; i++)
        if (!ctxs[i]) return TB_EINVAL;
    if (nframes == 0) return TB_OK;
    tb_ctx* c0 = ctxs[0];
    if (!left || !right || stride < p->width || pitch < (size_t)stride * p->height || p->nlevels < 2 || p->nlevels > TB_MAX_LEVELS)
        return tb_fail(c0, TB_EINVAL, "tb_batch_run: frame geometry / level count");
    std::vector<float> sf(p->nlevels), tmp(p->nlevels);
    tb_scale_factors(p->nlevels, p->scale, sf.data(), tmp.data(), tmp.data(), tmp.data());
    struct Shard {
        tb_ctx* ctx = nullptr;
        tb_extractor* ex = nullptr;
        int f0 = 0, m = 0;
        tb_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr; int32_t* d_counts = nullptr;   /* [2 m][cap] compact records */
        tb_match* d_matches = nullptr; int32_t* d_mcounts = nullptr;
    };
    std::vector<Shard> sh(ngpu);
    auto release = [&]() {
        for (Shard& s : sh) {
            if (!s.ctx) continue;
            hipSetDevice(s.ctx->device);
            hipStreamSynchronize(s.ctx->stream);
            if (s.ex) tb_extractor_destroy(s.ex);
            hipFree(s.d_kps); hipFree(s.d_desc); hipFree(s.d_counts); hipFree(s.d_matches); hipFree(s.d_mcounts);
        }
    };
    int rc = TB_OK;
    /* 1. queue every shard's chain: contiguous blocks of frames, left images [0, m), right images [m, 2 m) of the plan */
    for (int i = 0; i < ngpu && rc == TB_OK; i++) {
        Shard& s = sh[i];
        const int f0 = (int)((long long)nframes * i / ngpu), f1 = (int)((long long)nframes * (i + 1) / ngpu);
        if (f1 == f0) continue;
        s.ctx = ctxs[i]; s.f0 = f0; s.m = f1 - f0;
        tb_ctx* ctx = s.ctx;
        const int m = s.m;
        if ((rc = tb_extractor_create(ctx, p->width, p->height, p->nlevels, sf.data(), nullptr, nullptr, 2 * m, p->target, &s.ex))) break;
        tb_extractor* ex = s.ex;   /* tb_extractor_create has bound this thread to the context's device */
        const LevelGeom& L0 = ex->g.lv[0];
        hipError_t e = hipSuccess;
        for (int side = 0; side < 2 && e == hipSuccess; side++)
            for (int f = 0; f < m && e == hipSuccess; f++)
                e = hipMemcpy2DAsync(ex->d_slab + (size_t)(side * m + f) * ex->g.slabBytes + L0.off, L0.stride,
                                     (side ? right : left) + (size_t)(f0 + f) * pitch, stride, L0.w, L0.h, hipMemcpyHostToDevice,
                                     ctx->stream);
        if (e != hipSuccess) { rc = tb_fail(ctx, TB_EDEVICE, "tb_batch_run: frame upload: %s", hipGetErrorString(e)); break; }
        ex->g.img0 = nullptr;
        if ((rc = tb_extractor_build_pyramid(ex, 2 * m))) break;
        if ((rc = tb_extractor_orb(ex, 2 * m, p->target, p->init_th, p->min_th, 0, nullptr, 0))) break;
        if (hipMalloc(&s.d_kps, (size_t)2 * m * cap * sizeof(tb_keypoint)) != hipSuccess ||
            hipMalloc(&s.d_desc, (size_t)2 * m * cap * 32) != hipSuccess || hipMalloc(&s.d_counts, (size_t)2 * m * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(&s.d_matches, (size_t)m * cap * sizeof(tb_match)) != hipSuccess ||
            hipMalloc(&s.d_mcounts, (size_t)m * sizeof(int32_t)) != hipSuccess) {
            rc = tb_fail(ctx, TB_ENOMEM, "tb_batch_run: record buffers of shard %d", i);
            break;
        }
        if ((rc = tb_extractor_copy_results_dev(ex, 2 * m, s.d_kps, s.d_desc, s.d_counts, cap))) break;
        /* searchByBF on the plan's own descriptor sets: left set f against right set m + f */
        const uint8_t* dsc = nullptr; const int32_t* cnt = nullptr; int selCap = 0;
        tb_extractor_results_dev(ex, nullptr, &dsc, &cnt, &selCap);
        if ((rc = tb_search_by_bf_batch_dev(ctx, m, dsc, cnt, dsc + (size_t)m * selCap * 32, cnt + m, (size_t)selCap * 32, p->bf_ratio,
                                            p->bf_min_th, s.d_matches, cap, s.d_mcounts)))
            break;
    }
    /* 2. the exchange step: every shard's records into the caller's arrays (waits for that shard only; the others keep working) */
    for (int i = 0; i < ngpu && rc == TB_OK; i++) {
        Shard& s = sh[i];
        if (!s.ctx) continue;
        tb_ctx* ctx = s.ctx;
        const int m = s.m, f0 = s.f0;
        hipError_t e = hipSetDevice(ctx->device);
        for (int side = 0; side < 2 && e == hipSuccess; side++) {
            e = hipMemcpyAsync(kps + ((size_t)side * nframes + f0) * cap, s.d_kps + (size_t)side * m * cap, (size_t)m * cap * sizeof(tb_keypoint),
                               hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(desc + ((size_t)side * nframes + f0) * cap * 32, s.d_desc + (size_t)side * m * cap * 32, (size_t)m * cap * 32,
                                   hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(counts + (size_t)side * nframes + f0, s.d_counts + (size_t)side * m, (size_t)m * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(matches + (size_t)f0 * cap, s.d_matches, (size_t)m * cap * sizeof(tb_match), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(match_counts + f0, s.d_mcounts, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { rc = tb_fail(ctx, TB_EDEVICE, "tb_batch_run: gather of shard %d: %s", i, hipGetErrorString(e)); break; }
        /* the compact records were cut at cap: say so instead of handing back a truncated frame */
        std::vector<int32_t> full(2 * m);
        if (tb_extractor_counts_host(s.ex, 2 * m, full.data()) == TB_OK)
            for (int k = 0; k < 2 * m; k++)
                if (full[k] > cap) { rc = tb_fail(ctx, TB_ECAPACITY, "tb_batch_run: %d keypoints in a frame of shard %d, capacity %d", full[k], i, cap); break; }
    }
    release();
    return rc;
}

}  // extern "C"
