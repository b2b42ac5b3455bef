// Chebyshev graph convolution pieces (PyG ChebConv as used by model/model.py:53,96):
//   k_spmm  -- the CSR message-aggregate  out = alpha * L^ x + beta * p + gamma * q
//   k_gemm  -- tiled fp32 GEMM used for the gate GEMM  Y = [T_0 .. T_{K-1} | S] W, its
//              data gradient and (split over row blocks) its weight gradient
//   k_colsum -- fixed-order reduction of per-block partial sums
#include "qt_common.h"

namespace {

// ------------------------------------------------------------------ message aggregate
template <int VEC>
__global__ __launch_bounds__(256) void k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const float* __restrict__ nrm, int N, int C, const float* __restrict__ x,
                                              float alpha, const float* p, float beta, const float* q, float gamma,
                                              float* out) {  // out may alias p or q (in-place Clenshaw step)
    const int nch = C / VEC;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = idx / nch;
    if (row >= N) return;
    const int ch = (int)(idx % nch) * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0f;
    const int e1 = rowptr[row + 1];
    for (int e = rowptr[row]; e < e1; ++e) {
        const float w = nrm[e];
        const float* xp = x + (int64_t)col[e] * C + ch;
        if constexpr (VEC == 4) {
            const float4 f = *reinterpret_cast<const float4*>(xp);
            acc[0] += w * f.x; acc[1] += w * f.y; acc[2] += w * f.z; acc[3] += w * f.w;
        } else {
            acc[0] += w * xp[0];
        }
    }
    const int64_t o = row * C + ch;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float r = alpha * acc[k];
        if (p) r += beta * p[o + k];
        if (q) r += gamma * q[o + k];
        acc[k] = r;
    }
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        out[o] = acc[0];
    }
}

// ------------------------------------------------------------------ tiled GEMM
// Node-feature operand made of Ka planes (N, Ca) plus an optional (N, Ks) block.
struct PlaneSrc {
    const float* a0;
    const float* a_rest;
    const float* S;
    int Ka, Ca, Ks, N;
    __device__ __forceinline__ float at(int64_t row, int k) const {
        const int kc = Ka * Ca;
        if (k < kc) {
            const int pl = k / Ca, ch = k - pl * Ca;
            const float* base = pl == 0 ? a0 : a_rest + (int64_t)(pl - 1) * N * Ca;
            return base[row * Ca + ch];
        }
        return S[row * Ks + (k - kc)];
    }
    __device__ __forceinline__ int width() const { return Ka * Ca + Ks; }
};

struct GemmArgs {
    PlaneSrc A;        // forward: left operand rows = nodes; wgrad: transposed use
    const float* B;    // forward: W (K, NB); wgrad: G (N, NB)
    int M, K, NB;      // output M x NB, reduction K
    // forward epilogue
    int Kb, Cb, act;
    const float* res;
    int res_stride;
    const float* drop;
    float* out;
    int64_t row0_step;  // wgrad: rows per block
};

constexpr int TM = 64, TN = 64, TK = 16;

// MODE 0: out planes = act(A @ W);  MODE 1: part[blockIdx.z] = A[rows]^T @ G[rows]
template <int MODE>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    __shared__ float As[TK][TM + 4];
    __shared__ float Bs[TK][TN + 4];
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;
    const int i0 = blockIdx.x * TM, j0 = blockIdx.y * TN;
    int64_t kbeg = 0, kend = g.K;
    if (MODE == 1) {
        kbeg = (int64_t)blockIdx.z * g.row0_step;
        kend = min((int64_t)g.A.N, kbeg + g.row0_step);
    }
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;

    for (int64_t k0 = kbeg; k0 < kend; k0 += TK) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + 256 * u;
            if (MODE == 0) {
                const int i = e >> 4, k = e & 15;
                const int64_t row = i0 + i;
                As[k][i] = (row < g.M && k0 + k < kend) ? g.A.at(row, (int)(k0 + k)) : 0.0f;
            } else {
                const int i = e & 63, k = e >> 6;
                const int feat = i0 + i;
                As[k][i] = (feat < g.M && k0 + k < kend) ? g.A.at(k0 + k, feat) : 0.0f;
            }
            const int kb = e >> 6, j = e & 63;
            Bs[kb][j] = (j0 + j < g.NB && k0 + kb < kend) ? g.B[(k0 + kb) * g.NB + j0 + j] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float a_[4] = {av.x, av.y, av.z, av.w};
            const float b_[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(a_[a], b_[b], acc[a][b]);
        }
        __syncthreads();
    }

#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t i = i0 + ty * 4 + a;
        if (i >= g.M) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tx * 4 + b;
            if (j >= g.NB) continue;
            float r = acc[a][b];
            if (MODE == 0) {
                if (g.act == QT_ACT_RELU) r = fmaxf(r, 0.0f);
                if (g.act == QT_ACT_TANH_RES) r = tanhf((g.drop ? g.drop[i] : 1.0f) * r) + g.res[i * g.res_stride];
                const int pl = j / g.Cb, ch = j - pl * g.Cb;
                g.out[(int64_t)pl * g.M * g.Cb + i * g.Cb + ch] = r;
            } else {
                g.out[((int64_t)blockIdx.z * g.M + i) * g.NB + j] = r;
            }
        }
    }
}

__global__ void k_colsum(const float* __restrict__ part, int nblk, int64_t len, float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= len) return;
    float acc = 0.0f;
    for (int i = 0; i < nblk; ++i) acc += part[(int64_t)i * len + j];
    out[j] = acc;
}

constexpr int WGRAD_ROWS = 512;

}  // namespace

extern "C" int qt_spmm(const int32_t* rowptr, const int32_t* col, const float* nrm, int N, int C, const float* x,
                       float alpha, const float* p, float beta, const float* q, float gamma, float* out, void* stream) {
    QT_ARG(rowptr && col && nrm && x && out && C > 0, "bad arguments");
    QT_ARG(x != out, "out must not alias x");
    if (N <= 0) return QT_OK;
    const bool v4 = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)p | (uintptr_t)q) % 16) == 0);
    const int nch = v4 ? C / 4 : C;
    const int grid = qt_cdiv((int64_t)N * nch, 256);
    if (v4)
        hipLaunchKernelGGL(k_spmm<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, rowptr, col, nrm, N, C, x, alpha, p, beta, q, gamma, out);
    else
        hipLaunchKernelGGL(k_spmm<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, rowptr, col, nrm, N, C, x, alpha, p, beta, q, gamma, out);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_dense(const float* a0, const float* a_rest, int Ka, int Ca, const float* W, const float* S, int Ks,
                        const float* Ws, int Kb, int Cb, int N, int act, const float* res, int res_stride,
                        const float* drop, float* out, void* stream) {
    QT_ARG(a0 && W && out && Ka >= 1 && Ca >= 1 && Kb >= 1 && Cb >= 1, "bad arguments");
    QT_ARG(Ka == 1 || a_rest, "a_rest missing");
    QT_ARG((Ks == 0) || (S && Ws), "S / Ws missing");
    QT_ARG(Ks == 0 || Ws == W + (int64_t)Ka * Ca * Kb * Cb, "Ws must follow W contiguously ([W ; Ws] is one matrix)");
    QT_ARG(act == QT_ACT_NONE || Kb == 1, "activation needs Kb == 1");
    QT_ARG(act != QT_ACT_TANH_RES || res, "QT_ACT_TANH_RES needs res");
    if (N <= 0) return QT_OK;
    GemmArgs g;
    g.A.a0 = a0; g.A.a_rest = a_rest; g.A.S = S; g.A.Ka = Ka; g.A.Ca = Ca; g.A.Ks = Ks; g.A.N = N;
    g.B = W; g.M = N; g.K = Ka * Ca + Ks; g.NB = Kb * Cb;
    g.Kb = Kb; g.Cb = Cb; g.act = act; g.res = res; g.res_stride = res_stride; g.drop = drop; g.out = out; g.row0_step = 0;
    hipLaunchKernelGGL(k_gemm<0>, dim3(qt_cdiv(N, TM), qt_cdiv(g.NB, TN), 1), dim3(256), 0, (hipStream_t)stream, g);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_wgrad_blocks(int N) { return N > 0 ? qt_cdiv(N, WGRAD_ROWS) : 0; }

extern "C" int qt_wgrad(const float* a0, const float* a_rest, int Ka, int Ca, const float* S, int Ks, const float* G,
                        int Co, int N, float* part, void* stream) {
    QT_ARG(a0 && G && part && Ka >= 1 && Ca >= 1 && Co >= 1, "bad arguments");
    QT_ARG(Ka == 1 || a_rest, "a_rest missing");
    QT_ARG(Ks == 0 || S, "S missing");
    if (N <= 0) return QT_OK;
    GemmArgs g;
    g.A.a0 = a0; g.A.a_rest = a_rest; g.A.S = S; g.A.Ka = Ka; g.A.Ca = Ca; g.A.Ks = Ks; g.A.N = N;
    g.B = G; g.M = Ka * Ca + Ks; g.K = N; g.NB = Co;
    g.Kb = 1; g.Cb = Co; g.act = QT_ACT_NONE; g.res = nullptr; g.res_stride = 0; g.drop = nullptr; g.out = part;
    g.row0_step = WGRAD_ROWS;
    hipLaunchKernelGGL(k_gemm<1>, dim3(qt_cdiv(g.M, TM), qt_cdiv(Co, TN), qt_cdiv(N, WGRAD_ROWS)), dim3(256), 0,
                       (hipStream_t)stream, g);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_colsum(const float* part, int nblk, int64_t len, float* out, void* stream) {
    QT_ARG(part && out && len > 0 && nblk >= 0, "bad arguments");
    hipLaunchKernelGGL(k_colsum, dim3(qt_cdiv(len, 256)), dim3(256), 0, (hipStream_t)stream, part, nblk, len, out);
    QT_LAUNCHED();
    return QT_OK;
}
