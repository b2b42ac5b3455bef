// Clip-resident Chebyshev recurrences: ALL hops of one ChebConv pass in ONE launch, neighbour rows staged in LDS.
//
// The message aggregate  out = alpha L^ x + beta p + gamma q  (PyG ChebConv.propagate, model/model.py:53,96) is applied K - 1
// times in a row by every ChebConv stack: T_k = 2 L^ T_{k-1} - T_{k-2} forward, the Clenshaw recurrence backward.  As one
// launch per hop (k_spmm, cheb.hip) each hop is a grid-wide dependent chain index -> gather -> store through L2 / HBM, and
// at the benchmark's size (1.2e5 nodes) every launch is latency-bound at ~7.5 us for 24 MB.
//
// Two facts of the path make the whole recurrence local to a workgroup:
//   * the batched mesh is block diagonal -- a node's neighbours belong to its own clip, and a clip has at most n x m nodes
//     (4096 at 64 x 64);
//   * L^ acts on every channel separately, so a clip's (nodes x C) matrix splits into independent 4-channel column slices.
// One 1024-thread workgroup owns (clip c, slice s): a slice plane is 4096 rows x 16 B = 64 KB of LDS, two planes (T_{k-1}
// gathered, T_{k-2} read and overwritten in place by the owner of the row) are 128 KB of the CU's 160 KB.  The first four
// edges of each of a thread's four rows (the mesh build's ELL side array: column x4, weight x4) stay in registers for all
// hops, the edges beyond the fourth (rows of big cells beside small ones) come from a per-clip pool the mesh build wrote
// (qt_edges_norm: {local column, weight}, <= 4080 entries = the remaining 32 KB of LDS), the gathers are ds_read_b128, and
// between two hops there is one workgroup barrier -- waiting on LDS traffic only -- instead of a kernel boundary.
// HBM sees the operand once: Z read, K - 1 planes written (forward); K gradient planes read, one written (backward: the
// intermediate Clenshaw terms b_k never leave LDS).
//
// Arithmetic: the same fused multiply-adds in the same order as k_spmm (ELL slots 0..3, then the CSR tail of rows with more
// than four edges, then alpha * acc + beta * p + gamma * q), so the planes are bit-identical to the per-hop launches
// (tests/test_gpu_ops.py::test_clip_resident_recurrence_equals_per_hop_launches).
#include "qt_common.h"

namespace {

// Threads per workgroup and rows gathered together, measured at the bench shape (K = 5, C = 4 + 16, forward / backward, us per
// launch; tools/exp_clip.py): 1024 x 1 row: 21.4 / 21.3;  1024 x 2: 22.4 / 23.8;  512 x 2: 24.9 / 26.6;  512 x 4: 25.5 / 27.3 --
// sixteen waves that each wait on one row's five LDS reads beat eight waves with four rows in flight (per-hop launches: 29.6 / 31.4).
#ifndef QT_CLIP_T
#define QT_CLIP_T 1024
#endif
#ifndef QT_CLIP_GR
#define QT_CLIP_GR 1
#endif
constexpr int CL_ROWS = 4096;              // rows of a clip that fit: 2 planes x 4096 x 16 B = 128 KB
constexpr int CL_T = QT_CLIP_T;            // threads per workgroup
constexpr int CL_RPT = CL_ROWS / CL_T;     // rows per thread (8 registers per row for all hops: packed ELL columns, weights, tail
                                           // descriptor, row number; the backward 4 more for the prefetched A_k)
constexpr int CL_TAIL = QT_TAIL_CAP;       // LDS pool of tail edges (edges 5, 6, .. of a row) per clip: 8 B each, ~32 KB

struct ClipPart {
    const float* z;      // forward: T_0 slice source, (N, C) with row stride ld
    float* planes;       // forward: K - 1 output planes T_1 .., each SLICE-major (C / 4, Ncap, 4); backward: (K, Ncap, C) gradient planes
                         // (row-major), plane 0 rewritten
    int C, ld;
};

struct ClipArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* nrm;
    const int4* ell;
    const int32_t* node_off;       // (B + 1) first node of every clip (device): the valid rows of clip c are [off[c], off[c + 1])
    const int32_t* tail_cnt;       // (B * QT_TAIL_CNT_STRIDE) tail edges of every clip in the pool (may exceed the capacity: see tail_info)
    const int2* tail_pool;         // (B, CL_TAIL) {local column, weight bits}
    const uint32_t* tail_info;     // (N) per row: pool base | count << 16; 0 = at most four edges; base 0xffff = walk the CSR
    int B, K, nsa;                 // nsa: 4-channel slices of part a (part b's follow)
    int Ncap;                      // plane stride in rows (the capacity in static mode)
    int bwd_sm;                    // backward: the gradient planes 1 .. K-1 are slice-major (plane 0 is always row-major)
    ClipPart a, b;
#ifdef QT_CLIP_TIMING
    long long* dbg;                // diagnostics build (tools/exp_clip_timing.py): 16 stamps per workgroup
#endif
};
#ifdef QT_CLIP_TIMING
#define CL_STAMP(i) do { if (g.dbg && threadIdx.x == 0) g.dbg[(int64_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define CL_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Workgroup barrier between two hops: only LDS is shared between the threads, so the barrier waits for this wave's LDS
// operations (lgkmcnt) and NOT for its global stores / prefetches (vmcnt), which __syncthreads() would also drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS rows are addressed by BYTE offset inside a plane (row * 16 < 65536: two offsets per register)
__device__ __forceinline__ float4 lds_row(const char* plane, unsigned byte_off) { return *reinterpret_cast<const float4*>(plane + byte_off); }

// The tail of a row with more than four edges, from the LDS copy of the clip's pool.  Four pool entries and their four
// gathers are in flight per trip; the accumulation order is the CSR order, as in k_spmm.
__device__ __forceinline__ void gather_tail_lds(float4& a, const char* __restrict__ P, const int2* __restrict__ TE, unsigned info) {
    const unsigned base = info & 0xffffu, cnt = info >> 16;
    for (unsigned j0 = 0; j0 < cnt; j0 += 4) {
        int2 e[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) e[v] = TE[base + min(j0 + v, cnt - 1)];
        float4 f[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) f[v] = lds_row(P, (unsigned)e[v].x << 4);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if (j0 + v < cnt) {
                const float we = __int_as_float(e[v].y);
                a.x += we * f[v].x; a.y += we * f[v].y; a.z += we * f[v].z; a.w += we * f[v].w;
            }
        }
    }
}

// (pool full -- a clip with more than CL_TAIL tail edges: the row walks the CSR arrays instead; correct, slow, rare)
__device__ __forceinline__ void gather_tail_csr(float4& a, const char* __restrict__ P, unsigned row, int r0,
                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                const float* __restrict__ nrm) {
    const int e0 = rowptr[row] + 4, e1 = rowptr[row + 1];
    for (int e = e0; e < e1; ++e) {
        const unsigned cj = (unsigned)((col[e] - r0) & (CL_ROWS - 1)) << 4;
        const float we = nrm[e];
        const float4 f = lds_row(P, cj);
        a.x += we * f.x; a.y += we * f.y; a.z += we * f.z; a.w += we * f.w;
    }
}

template <bool BWD>
__global__ __launch_bounds__(CL_T) void k_cheb_clip(ClipArgs g) {
    __shared__ __attribute__((aligned(16))) char Pl[2 * CL_ROWS * 16];
    __shared__ int2 TE[CL_TAIL];
    const int t = threadIdx.x;
    const int c = (int)blockIdx.x % g.B, s = (int)blockIdx.x / g.B;
    const bool second = s >= g.nsa;
    const ClipPart& pt = second ? g.b : g.a;
    const int C = pt.C;
    const int ch = 4 * (second ? s - g.nsa : s);
    const int r0 = g.node_off[c];
    const int nr = min(g.node_off[c + 1] - r0, CL_ROWS);
    if (nr <= 0) return;                                   // (workgroup-uniform)
    const int ntail = min(g.tail_cnt[QT_TAIL_CNT_STRIDE * c], CL_TAIL);
    CL_STAMP(0);
    const int K = g.K;
    const unsigned pstride = (unsigned)g.Ncap * (unsigned)C;      // (K * Ncap * C < 2^31: checked by the host entry -- 32-bit offsets)

    // element offset of this slice's float4 of (gradient plane k, row): row-major (K, Ncap, C), or planes 1.. slice-major
    auto grad_off = [&](int k, unsigned row) -> unsigned {
        if (g.bwd_sm && k > 0) return (unsigned)k * pstride + ((unsigned)(ch >> 2) * (unsigned)g.Ncap + row) * 4u;
        return (unsigned)k * pstride + row * (unsigned)C + ch;
    };
    // Prologue, ONE memory phase: the first four edges of this thread's rows (kept in registers for every hop), the rows'
    // tail descriptors, their first operand and the clip's tail pool -- all requested before anything is used.  Rows past the
    // clip's count are clamped to its last row (valid loads, results discarded), so no load sits behind a branch.
    unsigned rowc[CL_RPT];
    int4 c4[CL_RPT], wb[CL_RPT];
    unsigned tinfo[CL_RPT];                 // per row: pool base | tail edge count << 16 (0: none; base 0xffff: walk the CSR)
    float4 first[CL_RPT];
    float4 nxt[CL_RPT];
#pragma unroll
    for (int u = 0; u < CL_RPT; ++u) {
        rowc[u] = (unsigned)(r0 + min(t + CL_T * u, nr - 1));
        c4[u] = g.ell[2 * rowc[u]];
        wb[u] = g.ell[2 * rowc[u] + 1];
        tinfo[u] = g.tail_info[rowc[u]];
        if constexpr (!BWD) {
            first[u] = ld4g(pt.z + (rowc[u] * (unsigned)pt.ld + ch));
        } else {
            first[u] = ld4g(pt.planes + grad_off(K - 1, rowc[u]));
        }
    }
    {
        const int2* src = g.tail_pool + (int64_t)c * CL_TAIL;
        int2 te[(CL_TAIL + CL_T - 1) / CL_T];
#pragma unroll
        for (int i = 0; i < (CL_TAIL + CL_T - 1) / CL_T; ++i)
            if (t + CL_T * i < ntail) te[i] = src[t + CL_T * i];
#pragma unroll
        for (int i = 0; i < (CL_TAIL + CL_T - 1) / CL_T; ++i)
            if (t + CL_T * i < ntail) TE[t + CL_T * i] = te[i];
    }
    CL_STAMP(1);
    unsigned lc[CL_RPT][2];
    float w[CL_RPT][4];
    bool tails = false;
#pragma unroll
    for (int u = 0; u < CL_RPT; ++u) {
        const bool ok = t + CL_T * u < nr;
        if (c4[u].w < 0) c4[u].w = ~c4[u].w;
        if (!ok) tinfo[u] = 0;
        tails |= tinfo[u] != 0;
        lc[u][0] = ((unsigned)((c4[u].x - r0) & (CL_ROWS - 1)) << 4) | ((unsigned)((c4[u].y - r0) & (CL_ROWS - 1)) << 20);
        lc[u][1] = ((unsigned)((c4[u].z - r0) & (CL_ROWS - 1)) << 4) | ((unsigned)((c4[u].w - r0) & (CL_ROWS - 1)) << 20);
        w[u][0] = __int_as_float(wb[u].x);
        w[u][1] = __int_as_float(wb[u].y);
        w[u][2] = __int_as_float(wb[u].z);
        w[u][3] = __int_as_float(wb[u].w);
        if (ok) *reinterpret_cast<float4*>(Pl + (t + CL_T * u) * 16) = first[u];
    }
    CL_STAMP(2);
    if constexpr (BWD) {       // A_{K-2} of the rows: needed at the end of the first hop's groups (requested here, not in the prologue:
#pragma unroll                 // its registers would sit beside the ELL vectors' and spill)
        for (int u = 0; u < CL_RPT; ++u) nxt[u] = ld4g(pt.planes + grad_off(K - 2, rowc[u]));
    }
    lds_barrier();
    CL_STAMP(3);
    int stamp = 4;
    (void)stamp;
    // One hop with the gathered plane at byte offset CO of Pl (compile-time: the hop loops below are unrolled by two, so the plane
    // offsets are instruction immediates).  GR rows at a time: their 4 GR gathers AND their own old values (OWN: the plane
    // being overwritten) are requested together, so a hop is CL_RPT / GR LDS round trips per wave, not two per row (one per gather group and
    // one per own-row read, as the first version had it: 3.6 us per hop with 0.7 us of gather work in it).  Then the rare tails,
    // then `fin(u, alpha * acc [+ beta * own])` stores row u.
    constexpr int GR = QT_CLIP_GR;
    constexpr unsigned PLANE = (unsigned)CL_ROWS * 16u;
    auto hop = [&](auto co_tag, auto own_tag, float alpha, float beta, auto&& addend, auto&& fin) {
        constexpr unsigned CO = decltype(co_tag)::value;
        constexpr bool OWN = decltype(own_tag)::value;
        const char* Pc = Pl + CO;
        char* Pn = Pl + (CO ^ PLANE);
        // (the packed column offsets and the row numbers are made opaque once per hop: otherwise the loop-invariant unpacked LDS
        // addresses -- one per gather -- and 64-bit row addresses are kept in registers across the hops, and the kernel spills)
#pragma unroll
        for (int u = 0; u < CL_RPT; ++u) asm volatile("" : "+v"(lc[u][0]), "+v"(lc[u][1]), "+v"(rowc[u]));
#pragma unroll
        for (int u0 = 0; u0 < CL_RPT; u0 += GR) {
            float4 f[GR][4], own[GR];
#pragma unroll
            for (int v = 0; v < GR; ++v) {
                const unsigned (&l)[2] = lc[u0 + v];
                f[v][0] = lds_row(Pc, l[0] & 0xffffu);
                f[v][1] = lds_row(Pc, l[0] >> 16);
                f[v][2] = lds_row(Pc, l[1] & 0xffffu);
                f[v][3] = lds_row(Pc, l[1] >> 16);
            }
            if constexpr (OWN) {
#pragma unroll
                for (int v = 0; v < GR; ++v) own[v] = lds_row(Pn, (unsigned)(t + CL_T * (u0 + v)) * 16u);
            }
            float4 acc[GR];
#pragma unroll
            for (int v = 0; v < GR; ++v) {
                const float (&ww)[4] = w[u0 + v];
                float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;      // (same fused multiply-adds in the same order as k_spmm)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ax += ww[e] * f[v][e].x; ay += ww[e] * f[v][e].y; az += ww[e] * f[v][e].z; aw += ww[e] * f[v][e].w;
                }
                acc[v] = make_float4(ax, ay, az, aw);
            }
            if (tails) {
#pragma unroll
                for (int v = 0; v < GR; ++v) {
                    if (tinfo[u0 + v]) {
                        if ((tinfo[u0 + v] & 0xffffu) != 0xffffu)
                            gather_tail_lds(acc[v], Pc, TE, tinfo[u0 + v]);
                        else
                            gather_tail_csr(acc[v], Pc, rowc[u0 + v], r0, g.rowptr, g.col, g.nrm);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < GR; ++v) {
                const int u = u0 + v;
                float4 r = make_float4(alpha * acc[v].x, alpha * acc[v].y, alpha * acc[v].z, alpha * acc[v].w);
                addend(u, r);                                  // (backward: + A_k of the row)
                if constexpr (OWN) {
                    r.x += beta * own[v].x; r.y += beta * own[v].y; r.z += beta * own[v].z; r.w += beta * own[v].w;
                }
                if (t + CL_T * u < nr) fin(u, r, reinterpret_cast<float4*>(Pn + (unsigned)(t + CL_T * u) * 16u));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        CL_STAMP(stamp++);
    };
    using Co0 = std::integral_constant<unsigned, 0u>;
    using Co1 = std::integral_constant<unsigned, PLANE>;
    using Yes = std::true_type;
    using No = std::false_type;
    auto none = [](int, float4&) {};
    if constexpr (!BWD) {
        // T_1 = L^ T_0;  T_k = 2 L^ T_{k-1} - T_{k-2}: the owner of a row reads its old value and overwrites it (nobody else reads
        // that plane during the hop)
        auto store = [&](int k) {
            // planes leave SLICE-major, (plane, 4-channel slice, N, 4): this workgroup's rows are contiguous, its stores coalesce
            // (row-major planes: 16-byte pieces 4 C bytes apart, 21.3 vs 19.5 us per K = 5 launch); the GEMMs that read the
            // planes reach a row's quad at slice base + 4 row (PlaneSrc.sm)
            float* outp = pt.planes + ((unsigned)(k - 1) * pstride + (unsigned)(ch >> 2) * (unsigned)g.Ncap * 4u);
            return [=, &rowc](int u, const float4& r, float4* own) {
                *own = r;
                *reinterpret_cast<float4*>(outp + rowc[u] * 4u) = r;
            };
        };
        hop(Co0{}, No{}, 1.0f, 0.0f, none, store(1));
        for (int k = 2; k < K; k += 2) {
            hop(Co1{}, Yes{}, 2.0f, -1.0f, none, store(k));
            if (k + 1 < K) hop(Co0{}, Yes{}, 2.0f, -1.0f, none, store(k + 1));
        }
    } else {
        // Clenshaw on the gradient planes A_0 .. A_{K-1}: b_{K-1} = A_{K-1}; b_k = A_k + 2 L^ b_{k+1} - b_{k+2};
        // out = A_0 + L^ b_1 - b_2 (written over A_0).  A_k is this thread's own row of plane k, requested during the hop before.
        auto add_ak = [&](int k) {
            return [=, &nxt, &rowc](int u, float4& r) {
                const float4 ak = nxt[u];                      // A_k of this row; its A_{k-1} is requested as soon as A_k is consumed
                r.x += 1.0f * ak.x; r.y += 1.0f * ak.y; r.z += 1.0f * ak.z; r.w += 1.0f * ak.w;
                if (k > 0) nxt[u] = ld4g(pt.planes + grad_off(k - 1, rowc[u]));
            };
        };
        // b_k stays in LDS; the last hop (k = 0) writes the result over A_0 in global memory
        auto put = [&](int k) {
            return [=, &rowc](int u, const float4& r, float4* own) {
                if (k == 0)
                    *reinterpret_cast<float4*>(pt.planes + (rowc[u] * (unsigned)C + ch)) = r;
                else
                    *own = r;
            };
        };
        int k = K - 2;
        hop(Co0{}, No{}, k == 0 ? 1.0f : 2.0f, 0.0f, add_ak(k), put(k));          // b_{K-2} = A_{K-2} + 2 L^ b_{K-1}
        for (--k; k >= 0; k -= 2) {
            hop(Co1{}, Yes{}, k == 0 ? 1.0f : 2.0f, -1.0f, add_ak(k), put(k));
            if (k - 1 >= 0) hop(Co0{}, Yes{}, k - 1 == 0 ? 1.0f : 2.0f, -1.0f, add_ak(k - 1), put(k - 1));
        }
    }
}

}  // namespace

extern "C" int qt_cheb_clip_rows(void) { return CL_ROWS; }
#ifdef QT_CLIP_TIMING
static long long* g_clip_dbg = nullptr;
extern "C" void qt_clip_timing_buffer(long long* p) { g_clip_dbg = p; }
#endif

struct ClipMesh {       // the mesh operands both entry points share
    const int32_t *rowptr, *col;
    const float* nrm;
    const int32_t *ell, *node_off, *tail_cnt, *tail_pool, *tail_info;
    int B;
};

static int clip_launch(bool bwd, const ClipMesh& m, int Ncap, int K, int Ca, const float* za, int lda, float* Pa, int Cb,
                       const float* zb, int ldb, float* Pb, void* stream, int bwd_sm = 0) {
    ClipArgs g;
    g.bwd_sm = bwd_sm != 0;
    g.rowptr = m.rowptr;
    g.col = m.col;
    g.nrm = m.nrm;
    g.ell = reinterpret_cast<const int4*>(m.ell);
    g.node_off = m.node_off;
    g.tail_cnt = m.tail_cnt;
    g.tail_pool = reinterpret_cast<const int2*>(m.tail_pool);
    g.tail_info = reinterpret_cast<const uint32_t*>(m.tail_info);
    g.B = m.B;
    g.K = K;
    g.nsa = Ca / 4;
    g.Ncap = Ncap;
#ifdef QT_CLIP_TIMING
    g.dbg = g_clip_dbg;
#endif
    g.a = ClipPart{za, Pa, Ca, lda ? lda : Ca};
    g.b = ClipPart{zb, Pb, Cb, ldb ? ldb : Cb};
    const int grid = m.B * (Ca / 4 + Cb / 4);
    if (bwd)
        hipLaunchKernelGGL(k_cheb_clip<true>, dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(k_cheb_clip<false>, dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    return 0;
}

#define CLIP_MESH_ARGS_OK                                                                                                        \
    QT_ARG(rowptr && col && nrm && ell && node_off && tail_cnt && tail_pool && tail_info && B > 0 && K >= 2,                    \
           "bad arguments (the ELL side array and the tail pool of qt_edges_norm are required)")

extern "C" int qt_cheb_clip_fwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* node_off, const int32_t* tail_cnt, const int32_t* tail_pool,
                                const int32_t* tail_info, int B, int N, int K, int Ca, const float* za, int lda, float* Ta, int Cb,
                                const float* zb, int ldb, float* Tb, void* stream) {
    CLIP_MESH_ARGS_OK;
    QT_ARG(za && Ta && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || (zb && Tb)), "bad operands");
    QT_ARG((lda | ldb) % 4 == 0, "row strides must be multiples of 4");
    QT_ARG((((uintptr_t)za | (uintptr_t)Ta | (uintptr_t)zb | (uintptr_t)Tb | (uintptr_t)ell | (uintptr_t)tail_pool) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31) && (int64_t)N * max(max(lda, ldb), 4) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    const ClipMesh m = {rowptr, col, nrm, ell, node_off, tail_cnt, tail_pool, tail_info, B};
    clip_launch(false, m, N, K, Ca, za, lda, Ta, Cb, zb, ldb, Tb, stream);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_cheb_clip_bwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* node_off, const int32_t* tail_cnt, const int32_t* tail_pool,
                                const int32_t* tail_info, int B, int N, int K, int Ca, float* Ga, int Cb, float* Gb, int planes_sm,
                                void* stream) {
    CLIP_MESH_ARGS_OK;
    QT_ARG(Ga && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || Gb), "bad operands");
    QT_ARG((((uintptr_t)Ga | (uintptr_t)Gb | (uintptr_t)ell | (uintptr_t)tail_pool) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    const ClipMesh m = {rowptr, col, nrm, ell, node_off, tail_cnt, tail_pool, tail_info, B};
    clip_launch(true, m, N, K, Ca, nullptr, 0, Ga, Cb, nullptr, 0, Gb, stream, planes_sm);
    QT_LAUNCHED();
    return QT_OK;
}
