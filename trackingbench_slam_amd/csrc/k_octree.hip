/* a5 -- DistributeOctTree: quadtree spatial distribution of FAST candidates to a per-level quota.
 *
 * Reference: ORBExtractor::DistributeOctTree + ExtractorNode::DivideNode
 * (src/extractors/ORBextractor.cpp:494-733, :416-491).  The reference is sequential (std::list, vector
 * copies per node, sort by (size, pointer)).  This is a level-synchronous restatement that produces the
 * SAME node list in the SAME order: one workgroup per (frame, level), keys never move -- each key only
 * carries the list position of its node -- and every reference step becomes scans over LDS tables:
 *
 *   divide step(D nodes in processing order): count keys per child quadrant (LDS atomics) ->
 *     children created in order n1..n4 per parent; std::list::push_front puts later children first, so
 *     new list = reverse(created children) ++ surviving nodes in old order; keys are re-pointed.
 *   BFS pass   = divide every node that holds > 1 key, in list order          (:565-624)
 *   sorted pass = divide nodes of the last pass by (size desc, creation desc)  (:635-696)
 *                 until the list reaches the quota: the stop index is a prefix-sum crossing.
 *   leaf pick  = max (response, first-in-candidate-order) per node via 64-bit LDS atomicMax (:703-716),
 *                exit-key 20-px veto (:718-729), output in list order.
 *
 * Deviations (as the CPU oracle): ties between equal-size nodes are broken by creation
 * order (later first) instead of by heap address; nIni < 1 is clamped to 1.
 * Bound: latency / LDS atomics, not HBM (a few KB of keys per level); see DESIGN.md.
 */
#include "tb_internal.h"
#include "tb_device.h"

#define OT_TMAX 1024 /* launched with 256 threads (default) or 1024 (large images: the per-key loops dominate) */

struct OtNode { short x0, y0, x1, y1; };

struct OtLds {
    OtNode* nb[2];
    int* cnt[2];
    unsigned short* flag[2]; /* bit0 noMore, bit1 created-last-step with > 1 keys */
    unsigned short* seq[2];
    int* slotBase;           /* per list position: 4*processing rank, or -1 */
    int* survScan;           /* per list position */
    unsigned short* parentPos;  /* per processing rank */
    unsigned short* parentPos2;
    int* cbase;              /* per processing rank: creation index of its first child */
    unsigned char* cmask;    /* per processing rank: non-empty children */
    int* childCnt;           /* 4 per processing rank; aliased by the sort keys and the leaf pick */
    int* tmp;                /* scan scratch */
};

__device__ __forceinline__ int ot_quadrant_i(const OtNode nd, int x, int y) {
    const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1); /* ceil(float(w)/2), ORBextractor.cpp:418 */
    const int my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
    return (x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3);
}
__device__ __forceinline__ int ot_quadrant_f(const OtNode nd, float x, float y) {
    const int mx = nd.x0 + ((nd.x1 - nd.x0 + 1) >> 1);
    const int my = nd.y0 + ((nd.y1 - nd.y0 + 1) >> 1);
    return (x < (float)mx) ? ((y < (float)my) ? 0 : 2) : ((y < (float)my) ? 1 : 3);
}
__device__ __forceinline__ OtNode ot_child(const OtNode p, int q) {
    const short mx = p.x0 + ((p.x1 - p.x0 + 1) >> 1);
    const short my = p.y0 + ((p.y1 - p.y0 + 1) >> 1);
    OtNode c;
    c.x0 = (q & 1) ? mx : p.x0;
    c.x1 = (q & 1) ? p.x1 : mx;
    c.y0 = (q & 2) ? my : p.y0;
    c.y1 = (q & 2) ? p.y1 : my;
    return c;
}

/* Divide the D nodes whose slotBase >= 0 (processing rank e = slotBase/4, parent position parentPos[e]).
 * cur = index of the current buffers. Returns new list length; *nToExpand = children with > 1 keys. */
__device__ int ot_divide(OtLds& S, int cur, int L, int D, const uint32_t* __restrict__ rec, uint32_t* __restrict__ knode,
                         int n, const float* __restrict__ exitk, int32_t* __restrict__ enode, int n_exit,
                         int* nToExpand, int* sh_counter) {
    const int tid = threadIdx.x, OT_T = blockDim.x;
    const OtNode* nb = S.nb[cur];
    for (int i = tid; i < 4 * D; i += OT_T) S.childCnt[i] = 0;
    if (tid == 0) *sh_counter = 0;
    __syncthreads();
    for (int k = tid; k < n; k += OT_T) {
        const int p = knode[k];
        const int sb = S.slotBase[p];
        if (sb >= 0) {
            const uint32_t r = rec[k];
            const int q = ot_quadrant_i(nb[p], r & 0xfff, (r >> 12) & 0xfff);
            atomicAdd(&S.childCnt[sb + q], 1);
            knode[k] = 0x80000000u | (uint32_t)(sb + q);
        }
    }
    for (int j = tid; j < n_exit; j += OT_T) {
        const int p = enode[j];
        if (p >= 0) {
            const int sb = S.slotBase[p];
            if (sb >= 0) enode[j] = 0x40000000 | (sb + ot_quadrant_f(nb[p], exitk[2 * j], exitk[2 * j + 1]));
        }
    }
    __syncthreads();
    for (int e = tid; e < D; e += OT_T) {
        int m = 0;
        for (int q = 0; q < 4; q++) m |= (S.childCnt[4 * e + q] > 0) << q;
        S.cmask[e] = (unsigned char)m;
        S.cbase[e] = __popc(m);
    }
    for (int p = tid; p < L; p += OT_T) S.survScan[p] = S.slotBase[p] < 0;
    __syncthreads();
    const int C = tb_block_excl_scan(S.cbase, D, S.tmp);
    const int Sv = tb_block_excl_scan(S.survScan, L, S.tmp);
    const int nxt = cur ^ 1;
    int expand = 0;
    for (int s = tid; s < 4 * D; s += OT_T) {
        const int c = S.childCnt[s];
        if (c > 0) {
            const int e = s >> 2, q = s & 3;
            const int ci = S.cbase[e] + __popc(S.cmask[e] & ((1 << q) - 1));
            const int pos = C - 1 - ci;
            S.nb[nxt][pos] = ot_child(nb[S.parentPos[e]], q);
            S.cnt[nxt][pos] = c;
            S.flag[nxt][pos] = (unsigned short)((c == 1 ? 1 : 0) | (c > 1 ? 2 : 0));
            S.seq[nxt][pos] = (unsigned short)ci;
            if (c > 1) expand++;
        }
    }
    for (int p = tid; p < L; p += OT_T) {
        if (S.slotBase[p] < 0) {
            const int pos = C + S.survScan[p];
            S.nb[nxt][pos] = nb[p];
            S.cnt[nxt][pos] = S.cnt[cur][p];
            S.flag[nxt][pos] = S.flag[cur][p] & 1;
            S.seq[nxt][pos] = 0;
        }
    }
    if (expand) atomicAdd(sh_counter, expand);
    for (int k = tid; k < n; k += OT_T) {
        const uint32_t v = knode[k];
        if (v & 0x80000000u) {
            const int s = v & 0x7fffffff, e = s >> 2, q = s & 3;
            knode[k] = C - 1 - (S.cbase[e] + __popc(S.cmask[e] & ((1 << q) - 1)));
        } else {
            knode[k] = C + S.survScan[v];
        }
    }
    for (int j = tid; j < n_exit; j += OT_T) {
        const int v = enode[j];
        if (v < 0) continue;
        if (v & 0x40000000) {
            const int s = v & 0x3fffffff, e = s >> 2, q = s & 3;
            enode[j] = (S.childCnt[s] > 0) ? (C - 1 - (S.cbase[e] + __popc(S.cmask[e] & ((1 << q) - 1)))) : -1;
        } else {
            enode[j] = C + S.survScan[v];
        }
    }
    __syncthreads();
    *nToExpand = *sh_counter;
    __syncthreads();
    return C + Sv;
}

__global__ void __launch_bounds__(OT_TMAX)
k_octree(PlanGeom g, const uint32_t* __restrict__ cand, const int32_t* __restrict__ candCount,
         uint32_t* __restrict__ knodeAll, const float* __restrict__ exitk, int n_exit, int32_t* __restrict__ enodeAll,
         uint32_t* __restrict__ sel, int32_t* __restrict__ selCount, int capMax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int level = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, OT_T = blockDim.x;
    const LevelGeom& G = g.lv[level];
    const int n = min(candCount[b * TB_MAX_LEVELS + level], G.candCap);
    const uint32_t* rec = cand + (size_t)b * g.candPerImage + G.candOff;
    uint32_t* knode = knodeAll + (size_t)b * g.candPerImage + G.candOff;
    int32_t* enode = enodeAll + ((size_t)b * g.nlevels + level) * (size_t)(n_exit > 0 ? n_exit : 1);
    uint32_t* out = sel + (size_t)b * g.selCap + G.selBase;
    if (n == 0 || G.nodeCap <= 0) {
        if (tid == 0) selCount[b * TB_MAX_LEVELS + level] = 0;
        return;
    }
    /* carve LDS (capMax entries per table; every offset a multiple of 16 bytes) */
    OtLds S;
    {
        unsigned char* p = smem;
        const size_t cap = (size_t)((capMax + 7) & ~7);
        S.nb[0] = (OtNode*)p; p += cap * 8;
        S.nb[1] = (OtNode*)p; p += cap * 8;
        S.cnt[0] = (int*)p; p += cap * 4;
        S.cnt[1] = (int*)p; p += cap * 4;
        S.slotBase = (int*)p; p += cap * 4;
        S.survScan = (int*)p; p += cap * 4;
        S.cbase = (int*)p; p += cap * 4;
        S.childCnt = (int*)p; p += cap * 16;
        S.flag[0] = (unsigned short*)p; p += cap * 2;
        S.flag[1] = (unsigned short*)p; p += cap * 2;
        S.seq[0] = (unsigned short*)p; p += cap * 2;
        S.seq[1] = (unsigned short*)p; p += cap * 2;
        S.parentPos = (unsigned short*)p; p += cap * 2;
        S.parentPos2 = (unsigned short*)p; p += cap * 2;
        S.cmask = (unsigned char*)p; p += cap;
        S.tmp = (int*)p;
    }
    /* all LDS lives in the dynamic region (keeps its base 16-byte aligned) */
    int& sh_counter = S.tmp[32]; /* behind the scan scratch (one int per wavefront + 1) */
    int& sh_val = S.tmp[33];
    const int N = G.quota;
    const int nIni = G.nIni;
    const float hX = G.hX;
    const int H = G.h - 2 * TB_BORDER;

    /* ---- initial nodes, ORBextractor.cpp:498-544 */
    for (int i = tid; i < nIni; i += OT_T) S.childCnt[i] = 0;
    __syncthreads();
    for (int k = tid; k < n; k += OT_T) {
        int bin = (int)TB_FDIV((float)(rec[k] & 0xfff), hX);
        bin = min(bin, nIni - 1);
        atomicAdd(&S.childCnt[bin], 1);
        knode[k] = bin;
    }
    __syncthreads();
    for (int i = tid; i < nIni; i += OT_T) S.survScan[i] = S.childCnt[i] > 0;
    __syncthreads();
    int L = tb_block_excl_scan(S.survScan, nIni, S.tmp);
    int cur = 0;
    for (int i = tid; i < nIni; i += OT_T) {
        const int c = S.childCnt[i];
        if (c > 0) {
            const int pos = S.survScan[i];
            OtNode nd;
            nd.x0 = (short)(int)TB_FMUL(hX, (float)i);
            nd.x1 = (short)(int)TB_FMUL(hX, (float)(i + 1));
            nd.y0 = 0;
            nd.y1 = (short)H;
            S.nb[0][pos] = nd;
            S.cnt[0][pos] = c;
            S.flag[0][pos] = (c == 1) ? 1 : 0;
            S.seq[0][pos] = 0;
        }
    }
    for (int k = tid; k < n; k += OT_T) knode[k] = S.survScan[knode[k]];
    for (int j = tid; j < n_exit; j += OT_T) enode[j] = (S.childCnt[0] > 0) ? 0 : -1;
    __syncthreads();

    /* ---- subdivision, ORBextractor.cpp:546-698 */
    bool finish = false;
    while (!finish) {
        const int prevSize = L;
        /* BFS pass: every node with more than one key, in list order */
        for (int p = tid; p < L; p += OT_T) S.slotBase[p] = (S.flag[cur][p] & 1) ? 0 : 1;
        __syncthreads();
        const int D = tb_block_excl_scan(S.slotBase, L, S.tmp);
        if (D == 0) break; /* lNodes.size()==prevSize */
        for (int p = tid; p < L; p += OT_T) {
            if (S.flag[cur][p] & 1) S.slotBase[p] = -1;
            else {
                const int e = S.slotBase[p];
                S.parentPos[e] = (unsigned short)p;
                S.slotBase[p] = 4 * e;
            }
        }
        __syncthreads();
        int nToExpand = 0;
        L = ot_divide(S, cur, L, D, rec, knode, n, exitk, enode, n_exit, &nToExpand, &sh_counter);
        cur ^= 1;
        if (L >= N || L == prevSize) {
            finish = true;
        } else if (L + nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = L;
                /* candidates = children of the previous step with > 1 keys */
                for (int p = tid; p < L; p += OT_T) S.slotBase[p] = (S.flag[cur][p] & 2) ? 1 : 0;
                __syncthreads();
                const int M = tb_block_excl_scan(S.slotBase, L, S.tmp);
                if (M == 0) break;
                for (int p = tid; p < L; p += OT_T) {
                    if (S.flag[cur][p] & 2) {
                        const int e = S.slotBase[p];
                        S.parentPos2[e] = (unsigned short)p;
                        S.slotBase[p] = 4 * e;
                    } else S.slotBase[p] = -1;
                }
                for (int i = tid; i < 4 * M; i += OT_T) S.childCnt[i] = 0;
                __syncthreads();
                for (int k = tid; k < n; k += OT_T) {
                    const int p = knode[k];
                    const int sb = S.slotBase[p];
                    if (sb >= 0) {
                        const uint32_t r = rec[k];
                        atomicAdd(&S.childCnt[sb + ot_quadrant_i(S.nb[cur][p], r & 0xfff, (r >> 12) & 0xfff)], 1);
                    }
                }
                __syncthreads();
                for (int e = tid; e < M; e += OT_T) {
                    int c = 0;
                    for (int q = 0; q < 4; q++) c += S.childCnt[4 * e + q] > 0;
                    S.cmask[e] = (unsigned char)c; /* number of children, for the prefix below */
                }
                __syncthreads();
                /* sort candidates by (size desc, creation desc): keys alias childCnt */
                unsigned long long* skey = (unsigned long long*)S.childCnt;
                int M2 = 1;
                while (M2 < M) M2 <<= 1;
                for (int e = tid; e < M2; e += OT_T) {
                    unsigned long long key = 0;
                    if (e < M) {
                        const int p = S.parentPos2[e];
                        key = ((unsigned long long)(unsigned)S.cnt[cur][p] << 32) |
                              ((unsigned long long)S.seq[cur][p] << 16) | (unsigned long long)e;
                    }
                    skey[e] = key;
                }
                __syncthreads();
                for (int k2 = 2; k2 <= M2; k2 <<= 1)
                    for (int j = k2 >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < M2; i += OT_T) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const unsigned long long a = skey[i], c = skey[ixj];
                                const bool desc = (i & k2) == 0;
                                if ((a < c) == desc) { skey[i] = c; skey[ixj] = a; }
                            }
                        }
                        __syncthreads();
                    }
                /* prefix of (children - 1) in processing order; first crossing of the quota */
                for (int j = tid; j < M; j += OT_T) S.cbase[j] = (int)S.cmask[(int)(skey[j] & 0xffff)] - 1;
                __syncthreads();
                tb_block_excl_scan(S.cbase, M, S.tmp);
                if (tid == 0) sh_val = M - 1;
                __syncthreads();
                for (int j = tid; j < M; j += OT_T) {
                    const int own = (int)S.cmask[(int)(skey[j] & 0xffff)] - 1;
                    const int before = prevSize2 + S.cbase[j];
                    if (before < N && before + own >= N) sh_val = j; /* unique: prefix is monotone */
                }
                __syncthreads();
                const int Dp = sh_val + 1;
                for (int p = tid; p < L; p += OT_T) S.slotBase[p] = -1;
                __syncthreads();
                for (int j = tid; j < Dp; j += OT_T) {
                    const int p = S.parentPos2[(int)(skey[j] & 0xffff)];
                    S.parentPos[j] = (unsigned short)p;
                    S.slotBase[p] = 4 * j;
                }
                __syncthreads();
                int dummy = 0;
                L = ot_divide(S, cur, L, Dp, rec, knode, n, exitk, enode, n_exit, &dummy, &sh_counter);
                cur ^= 1;
                if (L >= N || L == prevSize2) finish = true;
            }
            finish = true;
        }
    }

    /* ---- best key per leaf (max response, first in candidate order), ORBextractor.cpp:700-730 */
    unsigned long long* best = (unsigned long long*)S.childCnt;
    for (int p = tid; p < L; p += OT_T) { best[p] = 0; S.survScan[p] = 1; }
    __syncthreads();
    for (int k = tid; k < n; k += OT_T) {
        const uint32_t r = rec[k];
        const int x = r & 0xfff, y = (r >> 12) & 0xfff;
        const int ci = (y - 3) / G.hCell, cj = (x - 3) / G.wCell;
        const uint32_t ord = ((uint32_t)(ci * G.nCols + cj) << 12) | ((uint32_t)(y - ci * G.hCell) << 6) |
                             (uint32_t)(x - cj * G.wCell);
        const unsigned long long val = ((unsigned long long)(r >> 24) << 32) | (unsigned long long)(0xffffffffu - ord);
        atomicMax(&best[knode[k]], val);
    }
    __syncthreads();
    /* decode the winner's position from its order key */
    for (int p = tid; p < L; p += OT_T) {
        const unsigned long long bv = best[p];
        const uint32_t ord = 0xffffffffu - (uint32_t)bv;
        const int cc = (int)(ord >> 12), ci = cc / G.nCols, cj = cc - ci * G.nCols;
        const int y = ci * G.hCell + (int)((ord >> 6) & 63), x = cj * G.wCell + (int)(ord & 63);
        S.slotBase[p] = (int)(((uint32_t)(bv >> 32) << 24) | ((uint32_t)y << 12) | (uint32_t)x);
    }
    __syncthreads();
    for (int j = tid; j < n_exit; j += OT_T) {
        const int p = enode[j];
        if (p >= 0) {
            const uint32_t r = (uint32_t)S.slotBase[p];
            const float dx = TB_FSUB(exitk[2 * j], (float)(r & 0xfff));
            const float dy = TB_FSUB(exitk[2 * j + 1], (float)((r >> 12) & 0xfff));
            if (TB_FADD(TB_FMUL(dx, dx), TB_FMUL(dy, dy)) < 400.f) S.survScan[p] = 0;
        }
    }
    __syncthreads();
    for (int p = tid; p < L; p += OT_T) S.cbase[p] = S.survScan[p];
    __syncthreads();
    const int total = tb_block_excl_scan(S.cbase, L, S.tmp);
    for (int p = tid; p < L; p += OT_T)
        if (S.survScan[p] && S.cbase[p] < G.nodeCap) out[S.cbase[p]] = (uint32_t)S.slotBase[p];
    if (tid == 0) selCount[b * TB_MAX_LEVELS + level] = min(total, G.nodeCap);
}

static size_t ot_lds_bytes(int capMax) {
    const size_t cap = (size_t)((capMax + 7) & ~7);
    return cap * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 16 + 2 + 2 + 2 + 2 + 2 + 2 + 1) + 64 * sizeof(int);
}

int tbk_octree(tb_extractor* ex, int n, int n_exit) {
    tb_ctx* ctx = ex->ctx;
    int capMax = 8;
    for (int l = 0; l < ex->g.nlevels; l++) capMax = ex->g.lv[l].nodeCap > capMax ? ex->g.lv[l].nodeCap : capMax;
    if (capMax > TB_NODE_CAP_MAX)
        return tb_fail(ctx, TB_EUNSUPPORTED, "per-level quota %d exceeds the quadtree LDS capacity %d", capMax,
                       TB_NODE_CAP_MAX);
    const size_t lds = ot_lds_bytes(capMax);
    /* a function attribute belongs to the (function, device) pair: set per call on the context's device whenever the
     * launch needs more than the default 64 KB (no process-wide "done" flag) */
    if (lds > 64 * 1024)
        TB_HIP(ctx, hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    dim3 grid(ex->g.nlevels, n);
    /* one workgroup walks all candidates of its level once per divide step: on large images (candidates grow with the
     * pixel count) those loops, not the node tables, set the time, and four times the threads are worth the idle lanes
     * in the scans (3840x2160 / 8000 keypoints: 48 -> see DESIGN.md; 1280x720 stays at 256) */
    const int threads = (long long)ex->g.lv[0].w * ex->g.lv[0].h >= (1 << 21) ? OT_TMAX : 256;
    tb_prof_begin(ctx, "k_octree");
    hipLaunchKernelGGL(k_octree, grid, dim3(threads), lds, ctx->stream, ex->g, ex->d_cand, ex->d_candCount, ex->d_knode,
                       ex->d_exit, n_exit, ex->d_enode, ex->d_sel, ex->d_selCount, capMax);
    tb_prof_end(ctx);
    TB_HIP(ctx, hipGetLastError());
    return TB_OK;
}
