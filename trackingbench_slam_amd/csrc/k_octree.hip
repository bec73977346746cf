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
 *
 * Keys in registers: every divide step walks all candidates of the level twice (count, re-point). With the keys'
 * records and node positions in global memory (round 1) each walk was a chain of dependent HBM round trips -- the
 * kernel ran at memory LATENCY, 1.7 ms per 1024 images of 1280x720 with 216 MB of scratch traffic per 128 images. A
 * thread now keeps its keys (tid, tid + T, ...; up to KPT of them) in registers: record and node position, loaded
 * once, never stored. A level with more candidates than T x KPT (3840x2160 level 0) takes the global-memory form of
 * the same code (template parameter KPT = 0).
 */
#include "tb_internal.h"
#include "tb_device.h"

/* keys per thread held in registers (template parameter KPT; 0 = keys in global memory): 32 at 256 threads per block
 * (8192 candidates per level), 12 at 1024 threads (12288; the register budget of 16 wavefronts per block is 128) */
#define OT_KA(KPT) ((KPT) > 0 ? (KPT) : 1)
#define OT_TMAX 1024 /* launched with 256 threads (default) or 1024 (large images: the per-key loops dominate) */

struct OtNode { short x0, y0, x1, y1; };

struct OtLds {
    int T;                   /* threads at work on this level: the whole block, or ONE wavefront for a level whose keys fit
                                64 x KPT registers (the other wavefronts of the block have exited; no barrier waits for them) */
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

/* tb_block_excl_scan for the first T threads of the block (T a multiple of 64) */
__device__ inline int ot_scan(int* arr, int n, int* tmp, int T) {
    const int tid = threadIdx.x;
    const int per = (n + T - 1) / T;
    const int beg = min(tid * per, n), end = min(beg + per, n);
    int s = 0;
    for (int i = beg; i < end; i++) s += arr[i];
    const int incl = tb_wave_incl_scan_dpp(s);
    const int wave = tid >> 6, lane = tid & 63, nw = T >> 6;
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < nw; w++) { int t = tmp[w]; tmp[w] = acc; acc += t; }
        tmp[nw] = acc;
    }
    __syncthreads();
    int run = tmp[wave] + incl - s;
    for (int i = beg; i < end; i++) { int t = arr[i]; arr[i] = run; run += t; }
    const int total = tmp[nw];
    __syncthreads();
    return total;
}

/* Visit every key of the level: f(record, node position &). KPT > 0: the thread's keys live in the register arrays kr / kn
 * (key tid + i T in slot i); otherwise they are read from / written back to global memory. */
template <int KPT, class F>
__device__ __forceinline__ void ot_each(int T, int n, uint32_t (&kr)[OT_KA(KPT)], uint32_t (&kn)[OT_KA(KPT)], const uint32_t* __restrict__ rec,
                                        uint32_t* __restrict__ knode, F f) {
    const int tid = threadIdx.x;
    if constexpr (KPT > 0) {
#pragma unroll
        for (int i = 0; i < KPT; i++)
            if (tid + i * T < n) f(kr[i], kn[i]);
    } else {
        for (int k = tid; k < n; k += T) {
            uint32_t nd = knode[k];
            const uint32_t before = nd;
            f(rec[k], nd);
            if (nd != before) knode[k] = nd;
        }
    }
}

/* Divide the D nodes whose slotBase >= 0 (processing rank e = slotBase/4, parent position parentPos[e]).
 * cur = index of the current buffers. Returns new list length; *nToExpand = children with > 1 keys. */
template <int KPT>
__device__ __forceinline__ int ot_divide(OtLds& S, int cur, int L, int D, uint32_t (&kr)[OT_KA(KPT)], uint32_t (&kn)[OT_KA(KPT)],
                                         const uint32_t* __restrict__ rec, uint32_t* __restrict__ knode,
                                         int n, const float* __restrict__ exitk, int32_t* __restrict__ enode, int n_exit,
                                         int* nToExpand, int* sh_counter) {
    const int tid = threadIdx.x, OT_T = S.T;
    const OtNode* nb = S.nb[cur];
    for (int i = tid; i < 4 * D; i += OT_T) S.childCnt[i] = 0;
    if (tid == 0) *sh_counter = 0;
    __syncthreads();
    ot_each<KPT>(S.T, n, kr, kn, rec, knode, [&](uint32_t r, uint32_t& nd) {
        const int p = (int)nd;
        const int sb = S.slotBase[p];
        if (sb >= 0) {
            const int q = ot_quadrant_i(nb[p], r & 0xfff, (r >> 12) & 0xfff);
            atomicAdd(&S.childCnt[sb + q], 1);
            nd = 0x80000000u | (uint32_t)(sb + q);
        }
    });
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
    const int C = ot_scan(S.cbase, D, S.tmp, OT_T);
    const int Sv = ot_scan(S.survScan, L, S.tmp, OT_T);
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
    ot_each<KPT>(S.T, n, kr, kn, rec, knode, [&](uint32_t, uint32_t& nd) {
        const uint32_t v = nd;
        if (v & 0x80000000u) {
            const int s = v & 0x7fffffff, e = s >> 2, q = s & 3;
            nd = (uint32_t)(C - 1 - (S.cbase[e] + __popc(S.cmask[e] & ((1 << q) - 1))));
        } else {
            nd = (uint32_t)(C + S.survScan[v]);
        }
    });
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

template <int KPT>
__device__ __forceinline__ void ot_level(unsigned char* smem, const PlanGeom& g, const LevelGeom& G, int level, int b, int n,
                                         const uint32_t* __restrict__ rec, uint32_t* __restrict__ knode,
                                         const float* __restrict__ exitk, int n_exit, int32_t* __restrict__ enode,
                                         uint32_t* __restrict__ out, int32_t* __restrict__ selCount, int capMax, int OT_T) {
    const int tid = threadIdx.x;
    uint32_t kr[OT_KA(KPT)], kn[OT_KA(KPT)];
    if constexpr (KPT > 0) {
#pragma unroll
        for (int i = 0; i < KPT; i++) {
            const int k = tid + i * OT_T;
            kr[i] = k < n ? rec[k] : 0u;
            kn[i] = 0u;
        }
    }
    /* carve LDS (capMax entries per table; every offset a multiple of 16 bytes) */
    OtLds S;
    S.T = OT_T;
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
    if constexpr (KPT > 0) {
#pragma unroll
        for (int i = 0; i < KPT; i++)
            if (tid + i * OT_T < n) {
                int bin = (int)TB_FDIV((float)(kr[i] & 0xfff), hX);
                bin = min(bin, nIni - 1);
                atomicAdd(&S.childCnt[bin], 1);
                kn[i] = (uint32_t)bin;
            }
    } else {
        for (int k = tid; k < n; k += OT_T) {
            int bin = (int)TB_FDIV((float)(rec[k] & 0xfff), hX);
            bin = min(bin, nIni - 1);
            atomicAdd(&S.childCnt[bin], 1);
            knode[k] = bin;
        }
    }
    __syncthreads();
    for (int i = tid; i < nIni; i += OT_T) S.survScan[i] = S.childCnt[i] > 0;
    __syncthreads();
    int L = ot_scan(S.survScan, nIni, S.tmp, OT_T);
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
    ot_each<KPT>(S.T, n, kr, kn, rec, knode, [&](uint32_t, uint32_t& nd) { nd = (uint32_t)S.survScan[nd]; });
    for (int j = tid; j < n_exit; j += OT_T) enode[j] = (S.childCnt[0] > 0) ? 0 : -1;
    __syncthreads();

    /* ---- subdivision, ORBextractor.cpp:546-698 */
    bool finish = false;
    while (!finish) {
        const int prevSize = L;
        /* BFS pass: every node with more than one key, in list order */
        for (int p = tid; p < L; p += OT_T) S.slotBase[p] = (S.flag[cur][p] & 1) ? 0 : 1;
        __syncthreads();
        const int D = ot_scan(S.slotBase, L, S.tmp, OT_T);
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
        L = ot_divide<KPT>(S, cur, L, D, kr, kn, rec, knode, n, exitk, enode, n_exit, &nToExpand, &sh_counter);
        cur ^= 1;
        if (L >= N || L == prevSize) {
            finish = true;
        } else if (L + nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = L;
                /* candidates = children of the previous step with > 1 keys */
                for (int p = tid; p < L; p += OT_T) S.slotBase[p] = (S.flag[cur][p] & 2) ? 1 : 0;
                __syncthreads();
                const int M = ot_scan(S.slotBase, L, S.tmp, OT_T);
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
                ot_each<KPT>(S.T, n, kr, kn, rec, knode, [&](uint32_t r, uint32_t& nd) {
                    const int p = (int)nd;
                    const int sb = S.slotBase[p];
                    if (sb >= 0) atomicAdd(&S.childCnt[sb + ot_quadrant_i(S.nb[cur][p], r & 0xfff, (r >> 12) & 0xfff)], 1);
                });
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
                ot_scan(S.cbase, M, S.tmp, OT_T);
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
                L = ot_divide<KPT>(S, cur, L, Dp, kr, kn, rec, knode, n, exitk, enode, n_exit, &dummy, &sh_counter);
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
    ot_each<KPT>(S.T, n, kr, kn, rec, knode, [&](uint32_t r, uint32_t& nd) {
        const int x = r & 0xfff, y = (r >> 12) & 0xfff;
        const int ci = (y - 3) / G.hCell, cj = (x - 3) / G.wCell;
        const uint32_t ord = ((uint32_t)(ci * G.nCols + cj) << 12) | ((uint32_t)(y - ci * G.hCell) << 6) |
                             (uint32_t)(x - cj * G.wCell);
        const unsigned long long val = ((unsigned long long)(r >> 24) << 32) | (unsigned long long)(0xffffffffu - ord);
        atomicMax(&best[nd], val);
    });
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
    const int total = ot_scan(S.cbase, L, S.tmp, OT_T);
    for (int p = tid; p < L; p += OT_T)
        if (S.survScan[p] && S.cbase[p] < G.nodeCap) out[S.cbase[p]] = (uint32_t)S.slotBase[p];
    if (tid == 0) selCount[b * TB_MAX_LEVELS + level] = min(total, G.nodeCap);
}

#ifdef OT_TIMING   /* debug build: shader clocks per block, by level */
__device__ unsigned long long ot_times[64];
extern "C" int tb_debug_octree_times(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ot_times), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[64] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(ot_times), z, sizeof z); }
    return 0;
}
#endif

template <int TMAX, int KPT>
__global__ void __launch_bounds__(TMAX)
k_octree(PlanGeom g, const uint32_t* __restrict__ cand, const int32_t* __restrict__ candCount,
         uint32_t* __restrict__ knodeAll, const float* __restrict__ exitk, int n_exit, int32_t* __restrict__ enodeAll,
         uint32_t* __restrict__ sel, int32_t* __restrict__ selCount, int capMax, int level0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int level = level0 + blockIdx.x, b = blockIdx.y;
    const LevelGeom& G = g.lv[level];
    const int n = min(candCount[b * TB_MAX_LEVELS + level], G.candCap);
    const uint32_t* rec = cand + (size_t)b * g.candPerImage + G.candOff;
    uint32_t* knode = knodeAll + (size_t)b * g.candPerImage + G.candOff;
    int32_t* enode = enodeAll + ((size_t)b * g.nlevels + level) * (size_t)(n_exit > 0 ? n_exit : 1);
    uint32_t* out = sel + (size_t)b * g.selCap + G.selBase;
    if (n == 0 || G.nodeCap <= 0) {
        if (threadIdx.x == 0) selCount[b * TB_MAX_LEVELS + level] = 0;
        return;
    }
#ifdef OT_TIMING
    const unsigned long long t0_ = __builtin_readcyclecounter();
    struct Fin { unsigned long long t0; int level, n; __device__ ~Fin() { if (threadIdx.x == 0) { atomicAdd(&ot_times[level], __builtin_readcyclecounter() - t0); atomicAdd(&ot_times[16 + level], 1ull); atomicAdd(&ot_times[32 + level], (unsigned long long)n); } } } fin_{t0_, level, n};
#endif
    if (n <= 64 * KPT / 2 && TMAX > 64) {
        /* a small level: one wavefront holds every key, its "barriers" are waits on its own LDS traffic */
        if (threadIdx.x >= 64) return;
        ot_level<KPT>(smem, g, G, level, b, n, rec, knode, exitk, n_exit, enode, out, selCount, capMax, 64);
    } else if (n <= TMAX * KPT) {
        ot_level<KPT>(smem, g, G, level, b, n, rec, knode, exitk, n_exit, enode, out, selCount, capMax, TMAX);
    } else {
        ot_level<0>(smem, g, G, level, b, n, rec, knode, exitk, n_exit, enode, out, selCount, capMax, TMAX);
    }
}

static size_t ot_lds_bytes(int capMax) {
    const size_t cap = (size_t)((capMax + 7) & ~7);
    return cap * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 16 + 2 + 2 + 2 + 2 + 2 + 2 + 1) + 64 * sizeof(int);
}

int tbk_octree(tb_extractor* ex, int n, int n_exit) {
    tb_ctx* ctx = ex->ctx;
    const PlanGeom& g = ex->g;
    for (int l = 0; l < g.nlevels; l++)
        if (g.lv[l].nodeCap > TB_NODE_CAP_MAX)
            return tb_fail(ctx, TB_EUNSUPPORTED, "per-level quota %d exceeds the quadtree LDS capacity %d", g.lv[l].nodeCap,
                           TB_NODE_CAP_MAX);
    /* The kernel is bound by the latency of its many short, barrier-separated phases, i.e. by how many levels a CU works
     * on at once -- LDS (65 bytes per node-table entry) and registers (the keys) decide that. The pyramid's levels differ
     * by 25x in candidates and 5x in quota, so they go out in two launches: the large levels (at least 250 000 pixels:
     * levels 0-2 of 1280x720) as whole workgroups with node tables for the largest quota, the small ones as two
     * wavefronts per level (128 x 32 keys in registers; ONE wavefront if the level has at most 2048 candidates, the other
     * exits) with tables sized for their own largest quota. Either launch falls back to the global-memory key walk for a
     * level with more candidates than its registers hold. */
    const bool big = (long long)g.lv[0].w * g.lv[0].h >= (1 << 21);
    int nLarge = 0;
    while (nLarge < g.nlevels && (long long)g.lv[nLarge].w * g.lv[nLarge].h >= 250000) nLarge++;
    typedef void (*kern_t)(PlanGeom, const uint32_t*, const int32_t*, uint32_t*, const float*, int, int32_t*, uint32_t*, int32_t*, int, int);
    for (int part = 0; part < 2; part++) {
        const int l0 = part == 0 ? 0 : nLarge, l1 = part == 0 ? nLarge : g.nlevels;
        if (l1 <= l0) continue;
        int capMax = 8;
        for (int l = l0; l < l1; l++) capMax = g.lv[l].nodeCap > capMax ? g.lv[l].nodeCap : capMax;
        const size_t lds = ot_lds_bytes(capMax);
        /* one workgroup walks all candidates of its level once per divide step: on images of 2 MP and more those loops,
         * not the node tables, set the time, and 1024 threads are worth the idle lanes in the scans */
        const kern_t kern = part == 1 ? (kern_t)k_octree<128, 32> : big ? (kern_t)k_octree<OT_TMAX, 12> : (kern_t)k_octree<256, 32>;
        const int threads = part == 1 ? 128 : big ? OT_TMAX : 256;
        /* a function attribute belongs to the (function, device) pair: set per call on the context's device whenever the
         * launch needs more than the default 64 KB (no process-wide "done" flag) */
        if (lds > 64 * 1024)
            TB_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
        tb_prof_begin(ctx, "k_octree");
        hipLaunchKernelGGL(kern, dim3(l1 - l0, n), dim3(threads), lds, ctx->stream, g, ex->d_cand, ex->d_candCount, ex->d_knode,
                           ex->d_exit, n_exit, ex->d_enode, ex->d_sel, ex->d_selCount, capMax, l0);
        tb_prof_end(ctx);
        TB_HIP(ctx, hipGetLastError());
    }
    return TB_OK;
}
