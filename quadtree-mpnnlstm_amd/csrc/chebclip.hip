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
// hops, the gathers are ds_read_b128, and between two hops there is one workgroup barrier instead of a kernel boundary.
// HBM sees the operand once: Z read, K - 1 planes written (forward); K gradient planes read, one written (backward: the
// intermediate Clenshaw terms b_k never leave LDS).
//
// Arithmetic: the same fused multiply-adds in the same order as k_spmm (ELL slots 0..3, then the CSR tail of rows with more
// than four edges, then alpha * acc + beta * p + gamma * q), so the planes are bit-identical to the per-hop launches
// (tests/test_gpu_ops.py::test_clip_resident_recurrence_equals_per_hop_launches).
#include "qt_common.h"

namespace {

constexpr int CL_T = 1024;                 // threads per workgroup
constexpr int CL_RPT = 4;                  // rows per thread
constexpr int CL_ROWS = CL_T * CL_RPT;     // rows of a clip that fit: 2 planes x 4096 x 16 B = 128 KB

struct ClipPart {
    const float* z;      // forward: T_0 slice source, (N, C) with row stride ld
    float* planes;       // forward: (K - 1, Ncap, C) output planes T_1 ..; backward: (K, Ncap, C) gradient planes, plane 0 rewritten
    int C, ld;
};

struct ClipArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* nrm;
    const int4* ell;
    const int32_t* node_off;       // (B + 1) first node of every clip (device): the valid rows of clip c are [off[c], off[c + 1])
    int B, K, nsa;                 // nsa: 4-channel slices of part a (part b's follow)
    int Ncap;                      // plane stride in rows (the capacity in static mode)
    ClipPart a, b;
};

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }

// one row of  acc = sum_e nrm[e] * P[col[e]]  from the ELL slots in registers (+ the CSR tail for rows with more edges)
// (lc: the four local column indices, 12 bits each, packed two per register)
__device__ __forceinline__ float4 gather_row(const float4* __restrict__ P, const unsigned (&lc)[2], const float (&w)[4], int e0, int e1,
                                             int r0, const int32_t* __restrict__ col, const float* __restrict__ nrm) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
    const float4 f0 = P[lc[0] & 0xffffu], f1 = P[lc[0] >> 16], f2 = P[lc[1] & 0xffffu], f3 = P[lc[1] >> 16];
    ax += w[0] * f0.x; ay += w[0] * f0.y; az += w[0] * f0.z; aw += w[0] * f0.w;
    ax += w[1] * f1.x; ay += w[1] * f1.y; az += w[1] * f1.z; aw += w[1] * f1.w;
    ax += w[2] * f2.x; ay += w[2] * f2.y; az += w[2] * f2.z; aw += w[2] * f2.w;
    ax += w[3] * f3.x; ay += w[3] * f3.y; az += w[3] * f3.z; aw += w[3] * f3.w;
    for (int e = e0; e < e1; ++e) {                       // rows with more than four edges (a big cell beside small ones)
        const int cj = (col[e] - r0) & (CL_ROWS - 1);
        const float we = nrm[e];
        const float4 f = P[cj];
        ax += we * f.x; ay += we * f.y; az += we * f.z; aw += we * f.w;
    }
    return make_float4(ax, ay, az, aw);
}

template <bool BWD>
__global__ __launch_bounds__(CL_T) void k_cheb_clip(ClipArgs g) {
    __shared__ float4 P[2][CL_ROWS];
    const int t = threadIdx.x;
    const int c = (int)blockIdx.x % g.B, s = (int)blockIdx.x / g.B;
    const bool second = s >= g.nsa;
    const ClipPart& pt = second ? g.b : g.a;
    const int C = pt.C;
    const int ch = 4 * (second ? s - g.nsa : s);
    const int r0 = g.node_off[c];
    const int nr = min(g.node_off[c + 1] - r0, CL_ROWS);
    if (nr <= 0) return;                                   // (workgroup-uniform)

    // the first four edges of this thread's rows: local column indices and weights, kept for every hop
    unsigned lc[CL_RPT][2];
    float w[CL_RPT][4];
    int e0[CL_RPT], e1[CL_RPT];
    bool ok[CL_RPT];
#pragma unroll
    for (int u = 0; u < CL_RPT; ++u) {
        const int lr = t + CL_T * u;
        ok[u] = lr < nr;
        e0[u] = e1[u] = 0;
        lc[u][0] = lc[u][1] = 0u;
#pragma unroll
        for (int v = 0; v < 4; ++v) w[u][v] = 0.0f;
        if (ok[u]) {
            const unsigned row = (unsigned)(r0 + lr);
            int4 c4 = g.ell[2 * row];
            const int4 wb = g.ell[2 * row + 1];
            if (c4.w < 0) {
                c4.w = ~c4.w;
                e0[u] = g.rowptr[row] + 4;
                e1[u] = g.rowptr[row + 1];
            }
            lc[u][0] = (unsigned)((c4.x - r0) & (CL_ROWS - 1)) | ((unsigned)((c4.y - r0) & (CL_ROWS - 1)) << 16);
            lc[u][1] = (unsigned)((c4.z - r0) & (CL_ROWS - 1)) | ((unsigned)((c4.w - r0) & (CL_ROWS - 1)) << 16);
            w[u][0] = __int_as_float(wb.x);
            w[u][1] = __int_as_float(wb.y);
            w[u][2] = __int_as_float(wb.z);
            w[u][3] = __int_as_float(wb.w);
        }
    }
    const int K = g.K;
    const unsigned pstride = (unsigned)g.Ncap * (unsigned)C;      // (K * Ncap * C < 2^31: checked by the host entry -- 32-bit offsets)
    int cur = 0;
    if constexpr (!BWD) {
        // T_0 slice -> plane 0
#pragma unroll
        for (int u = 0; u < CL_RPT; ++u)
            if (ok[u]) P[0][t + CL_T * u] = ld4g(pt.z + ((unsigned)(r0 + t + CL_T * u) * (unsigned)pt.ld + ch));
        __syncthreads();
        for (int k = 1; k < K; ++k) {
            const float alpha = k == 1 ? 1.0f : 2.0f;
            float* outp = pt.planes + ((unsigned)(k - 1) * pstride + ch);
#pragma unroll
            for (int u = 0; u < CL_RPT; ++u) {
                if (!ok[u]) continue;
                const int lr = t + CL_T * u;
                const float4 a = gather_row(P[cur], lc[u], w[u], e0[u], e1[u], r0, g.col, g.nrm);
                float4 r = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
                if (k > 1) {                                   // T_k = 2 L^ T_{k-1} - T_{k-2}: the owner of a row reads its old value
                    const float4 pv = P[cur ^ 1][lr];
                    r.x += -1.0f * pv.x; r.y += -1.0f * pv.y; r.z += -1.0f * pv.z; r.w += -1.0f * pv.w;
                }
                P[cur ^ 1][lr] = r;                            // (and overwrites it: nobody else reads that plane in this hop)
                *reinterpret_cast<float4*>(outp + (unsigned)(r0 + lr) * (unsigned)C) = r;
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // Clenshaw on the gradient planes A_0 .. A_{K-1}: b_{K-1} = A_{K-1}; b_k = A_k + 2 L^ b_{k+1} - b_{k+2};
        // out = A_0 + L^ b_1 - b_2 (written over A_0).  A_k is this thread's own row of plane k, requested one hop ahead.
        float4 nxt[CL_RPT];
#pragma unroll
        for (int u = 0; u < CL_RPT; ++u) {
            nxt[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) {
                const unsigned off = (unsigned)(r0 + t + CL_T * u) * (unsigned)C + ch;
                P[0][t + CL_T * u] = ld4g(pt.planes + ((unsigned)(K - 1) * pstride + off));
                nxt[u] = ld4g(pt.planes + ((unsigned)(K - 2) * pstride + off));
            }
        }
        __syncthreads();
        for (int k = K - 2; k >= 0; --k) {
            const float alpha = k == 0 ? 1.0f : 2.0f;
#pragma unroll
            for (int u = 0; u < CL_RPT; ++u) {
                if (!ok[u]) continue;
                const float4 ak = nxt[u];                      // A_k; the row's A_{k-1} is requested before the gathers
                if (k > 0) nxt[u] = ld4g(pt.planes + ((unsigned)(k - 1) * pstride + (unsigned)(r0 + t + CL_T * u) * (unsigned)C + ch));
                const int lr = t + CL_T * u;
                const float4 a = gather_row(P[cur], lc[u], w[u], e0[u], e1[u], r0, g.col, g.nrm);
                float4 r = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
                r.x += 1.0f * ak.x; r.y += 1.0f * ak.y; r.z += 1.0f * ak.z; r.w += 1.0f * ak.w;
                if (k + 2 < K) {
                    const float4 qv = P[cur ^ 1][lr];
                    r.x += -1.0f * qv.x; r.y += -1.0f * qv.y; r.z += -1.0f * qv.z; r.w += -1.0f * qv.w;
                }
                if (k == 0)
                    *reinterpret_cast<float4*>(pt.planes + ((unsigned)(r0 + lr) * (unsigned)C + ch)) = r;
                else
                    P[cur ^ 1][lr] = r;
            }
            __syncthreads();
            cur ^= 1;
        }
    }
}

}  // namespace

extern "C" int qt_cheb_clip_rows(void) { return CL_ROWS; }

static int clip_launch(bool bwd, const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                       const int32_t* node_off, int B, int Ncap, int K, int Ca, const float* za, int lda, float* Pa, int Cb,
                       const float* zb, int ldb, float* Pb, void* stream) {
    ClipArgs g;
    g.rowptr = rowptr;
    g.col = col;
    g.nrm = nrm;
    g.ell = reinterpret_cast<const int4*>(ell);
    g.node_off = node_off;
    g.B = B;
    g.K = K;
    g.nsa = Ca / 4;
    g.Ncap = Ncap;
    g.a = ClipPart{za, Pa, Ca, lda ? lda : Ca};
    g.b = ClipPart{zb, Pb, Cb, ldb ? ldb : Cb};
    const int grid = B * (Ca / 4 + Cb / 4);
    if (bwd)
        hipLaunchKernelGGL(k_cheb_clip<true>, dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL(k_cheb_clip<false>, dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    return 0;
}

extern "C" int qt_cheb_clip_fwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* node_off, int B, int N, int K, int Ca, const float* za, int lda, float* Ta, int Cb,
                                const float* zb, int ldb, float* Tb, void* stream) {
    QT_ARG(rowptr && col && nrm && ell && node_off && B > 0 && K >= 2, "bad arguments (the ELL side array is required)");
    QT_ARG(za && Ta && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || (zb && Tb)), "bad operands");
    QT_ARG((lda | ldb) % 4 == 0, "row strides must be multiples of 4");
    QT_ARG((((uintptr_t)za | (uintptr_t)Ta | (uintptr_t)zb | (uintptr_t)Tb | (uintptr_t)ell) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31) && (int64_t)N * max(max(lda, ldb), 4) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    clip_launch(false, rowptr, col, nrm, ell, node_off, B, N, K, Ca, za, lda, Ta, Cb, zb, ldb, Tb, stream);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_cheb_clip_bwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* node_off, int B, int N, int K, int Ca, float* Ga, int Cb, float* Gb, void* stream) {
    QT_ARG(rowptr && col && nrm && ell && node_off && B > 0 && K >= 2, "bad arguments (the ELL side array is required)");
    QT_ARG(Ga && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || Gb), "bad operands");
    QT_ARG((((uintptr_t)Ga | (uintptr_t)Gb | (uintptr_t)ell) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    clip_launch(true, rowptr, col, nrm, ell, node_off, B, N, K, Ca, nullptr, 0, Ga, Cb, nullptr, 0, Gb, stream);
    QT_LAUNCHED();
    return QT_OK;
}
