// Gate-weight packing of a GConvLSTM whose eight GraphConv stacks hold TWO ChebConvs each (model/model.py:59-97, no
// nonlinearity in between, :95-96): the stacks are composed in weight space into one Chebyshev series of order 2K-1,
//   T_a T_b = (T_{a+b} + T_|a-b|) / 2   =>   M[k] = sum_{a,b} cf(k,a,b) W0[a] W1[b],
// and written straight into the packed gate matrix W ((2K-1) C + pad4(K), 4h) that k_gemm_fwd multiplies [T_k(L^) Z | T_k(L^) 1]
// with (Z = [X | H], columns = gate-major i, f, c, o).  The first layer's bias rides through the second layer as a
// series of its own (orders 0 .. K-1) and lands in the bias rows next to the second layer's bias.
//
// Parameter-sized work (a few thousand outputs of a few hundred FMAs): one launch forward, one backward, instead of the
// ~110 bmm / pad / cat / slice kernels the same algebra costs through torch ops and their autograd (0.3 ms per step).
#include "qt_common.h"

namespace {

struct Branch {
    const float* P0;   // (4, K, in, h)   first-layer lins^T per gate
    const float* B0;   // (4, h)
    const float* P1;   // (4, K, h, h)    second layer
    const float* B1;   // (4, h)
};

struct ComposeArgs {
    Branch x, hb;
    int K, cin, cin_pad, h;
    float* W1;         // variant with H: ((2K-1)(cin_pad + h) + ksp, 4h), or NULL
    float* W0;         // variant without H: ((2K-1) cin_pad + ksp, 4h), or NULL
    float* WT1;        // the transposes (4h, rows), optional: what the gate GEMM stages its weight chunk from
    float* WT0;
    // backward
    const float* gW1;
    const float* gW0;
    float *gxP0, *gxB0, *gxP1, *gxB1, *ghP0, *ghB0, *ghP1, *ghB1;
};

__device__ __forceinline__ float cf(int k, int a, int b) {
    return 0.5f * (float)((a + b == k) + ((a > b ? a - b : b - a) == k));
}

// M[g, k, c, p] of one branch
__device__ float series_elem(const Branch& br, int g, int k, int c, int p, int in, int K, int h) {
    float s = 0.0f;
    for (int a = 0; a < K; ++a)
        for (int b = 0; b < K; ++b) {
            const float f = cf(k, a, b);
            if (f == 0.0f) continue;
            const float* r0 = br.P0 + ((int64_t)(g * K + a) * in + c) * h;
            const float* r1 = br.P1 + (int64_t)(g * K + b) * h * h + p;
            float d = 0.0f;
            for (int o = 0; o < h; ++o) d = fmaf(r0[o], r1[(int64_t)o * h], d);
            s = fmaf(f, d, s);
        }
    return s;
}

// bias series order k (< K) of one branch: B0 through W1[k], plus B1 at order 0
__device__ float bias_elem(const Branch& br, int g, int k, int p, int K, int h) {
    const float* r1 = br.P1 + (int64_t)(g * K + k) * h * h + p;
    float d = 0.0f;
    for (int o = 0; o < h; ++o) d = fmaf(br.B0[g * h + o], r1[(int64_t)o * h], d);
    return k == 0 ? d + br.B1[g * h + p] : d;
}

__global__ __launch_bounds__(256) void k_compose2_fwd(ComposeArgs A) {
    const int K = A.K, K2 = 2 * K - 1, h = A.h, nc = 4 * h, ksp = (K + 3) / 4 * 4;
    const int C1 = A.cin_pad + h, C0 = A.cin_pad;
    const int n1 = A.W1 ? (K2 * C1 + ksp) * nc : 0, n0 = A.W0 ? (K2 * C0 + ksp) * nc : 0;
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n1 + n0) return;
    const bool with_h = idx < n1;
    if (!with_h) idx -= n1;
    const int C = with_h ? C1 : C0;
    float* W = with_h ? A.W1 : A.W0;
    float* WT = with_h ? A.WT1 : A.WT0;
    const int rows = K2 * C + ksp;
    const int row = idx / nc, col = idx - row * nc;
    const int g = col / h, p = col - g * h;
    float v = 0.0f;
    if (row < K2 * C) {
        const int k = row / C, c = row - k * C;
        if (c < A.cin)
            v = series_elem(A.x, g, k, c, p, A.cin, K, h);
        else if (c >= A.cin_pad)
            v = series_elem(A.hb, g, k, c - A.cin_pad, p, h, K, h);
    } else {
        const int k = row - K2 * C;
        if (k < K) v = bias_elem(A.x, g, k, p, K, h) + bias_elem(A.hb, g, k, p, K, h);
    }
    W[idx] = v;
    if (WT) WT[(int64_t)col * rows + row] = v;
}

// dL/dM[g, k, c, p] of the x branch (both variants) / the h branch, and of the bias rows
__device__ __forceinline__ float gw_x(const ComposeArgs& A, int k, int c, int col) {
    const int nc = 4 * A.h;
    float v = 0.0f;
    if (A.gW1) v += A.gW1[(int64_t)(k * (A.cin_pad + A.h) + c) * nc + col];
    if (A.gW0) v += A.gW0[(int64_t)(k * A.cin_pad + c) * nc + col];
    return v;
}
__device__ __forceinline__ float gw_h(const ComposeArgs& A, int k, int c, int col) {
    return A.gW1 ? A.gW1[(int64_t)(k * (A.cin_pad + A.h) + A.cin_pad + c) * (4 * A.h) + col] : 0.0f;
}
__device__ __forceinline__ float gw_b(const ComposeArgs& A, int k, int col) {
    const int nc = 4 * A.h, K2 = 2 * A.K - 1;
    float v = 0.0f;
    if (A.gW1) v += A.gW1[(int64_t)(K2 * (A.cin_pad + A.h) + k) * nc + col];
    if (A.gW0) v += A.gW0[(int64_t)(K2 * A.cin_pad + k) * nc + col];
    return v;
}

template <bool HB>
__device__ __forceinline__ float gw_m(const ComposeArgs& A, int k, int c, int col) {
    return HB ? gw_h(A, k, c, col) : gw_x(A, k, c, col);
}

template <bool HB>
__device__ void branch_bwd(const ComposeArgs& A, int idx) {
    const Branch& br = HB ? A.hb : A.x;
    const int K = A.K, K2 = 2 * K - 1, h = A.h;
    const int in = HB ? h : A.cin;
    float* gP0 = HB ? A.ghP0 : A.gxP0;
    float* gB0 = HB ? A.ghB0 : A.gxB0;
    float* gP1 = HB ? A.ghP1 : A.gxP1;
    float* gB1 = HB ? A.ghB1 : A.gxB1;
    const int nP0 = 4 * K * in * h, nP1 = 4 * K * h * h, nB = 4 * h;
    if (idx < nP0) {                         // gP0[g, a, c, o] = sum_{b, k} cf sum_p gM[g, k, c, p] P1[g, b, o, p]
        const int o = idx % h, c = (idx / h) % in, a = (idx / (h * in)) % K, g = idx / (h * in * K);
        float s = 0.0f;
        for (int b = 0; b < K; ++b)
            for (int k = 0; k < K2; ++k) {
                const float f = cf(k, a, b);
                if (f == 0.0f) continue;
                const float* r1 = br.P1 + ((int64_t)(g * K + b) * h + o) * h;
                float d = 0.0f;
                for (int p = 0; p < h; ++p) d = fmaf(gw_m<HB>(A, k, c, g * h + p), r1[p], d);
                s = fmaf(f, d, s);
            }
        gP0[idx] = s;
        return;
    }
    idx -= nP0;
    if (idx < nP1) {                         // gP1[g, b, o, p] = sum_{a, k} cf sum_c P0[g, a, c, o] gM[g, k, c, p] + B0[g, o] gBias[b, g, p]
        const int p = idx % h, o = (idx / h) % h, b = (idx / (h * h)) % K, g = idx / (h * h * K);
        float s = 0.0f;
        for (int a = 0; a < K; ++a)
            for (int k = 0; k < K2; ++k) {
                const float f = cf(k, a, b);
                if (f == 0.0f) continue;
                const float* r0 = br.P0 + (int64_t)(g * K + a) * in * h + o;
                float d = 0.0f;
                for (int c = 0; c < in; ++c) d = fmaf(r0[(int64_t)c * h], gw_m<HB>(A, k, c, g * h + p), d);
                s = fmaf(f, d, s);
            }
        gP1[idx] = fmaf(br.B0[g * h + o], gw_b(A, b, g * h + p), s);
        return;
    }
    idx -= nP1;
    if (idx < nB) {                          // gB0[g, o] = sum_{k < K, p} gBias[k, g, p] P1[g, k, o, p]
        const int o = idx % h, g = idx / h;
        float s = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float* r1 = br.P1 + ((int64_t)(g * K + k) * h + o) * h;
            for (int p = 0; p < h; ++p) s = fmaf(gw_b(A, k, g * h + p), r1[p], s);
        }
        gB0[idx] = s;
        return;
    }
    idx -= nB;
    if (idx < nB) gB1[idx] = gw_b(A, 0, idx);
}

__global__ __launch_bounds__(256) void k_compose2_bwd(ComposeArgs A) {
    const int K = A.K, h = A.h;
    const int nx = 4 * K * A.cin * h + 4 * K * h * h + 8 * h, nh = 8 * K * h * h + 8 * h;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < nx)
        branch_bwd<false>(A, idx);
    else if (idx < nx + nh)
        branch_bwd<true>(A, idx - nx);
}

int check(const char* fn, const ComposeArgs& A) {
    const Branch* b[2] = {&A.x, &A.hb};
    for (int i = 0; i < 2; ++i)
        if (!b[i]->P0 || !b[i]->B0 || !b[i]->P1 || !b[i]->B1) {
            qt_set_error("%s: null weight pointer", fn);
            return QT_E_ARG;
        }
    if (A.K < 1 || A.K > 8 || A.h < 1 || A.cin < 1 || A.cin_pad < A.cin || A.cin_pad % 4 || A.h % 4) {
        qt_set_error("%s: bad sizes (K in 1..8, cin <= cin_pad, cin_pad and h multiples of 4)", fn);
        return QT_E_ARG;
    }
    return QT_OK;
}

}  // namespace

extern "C" int qt_compose2_fwd(const float* Px0, const float* Bx0, const float* Px1, const float* Bx1, const float* Ph0,
                               const float* Bh0, const float* Ph1, const float* Bh1, int K, int cin, int cin_pad, int h,
                               float* W1, float* W0, float* WT1, float* WT0, void* stream) {
    ComposeArgs A = {};
    A.x = {Px0, Bx0, Px1, Bx1};
    A.hb = {Ph0, Bh0, Ph1, Bh1};
    A.K = K; A.cin = cin; A.cin_pad = cin_pad; A.h = h; A.W1 = W1; A.W0 = W0; A.WT1 = W1 ? WT1 : nullptr; A.WT0 = W0 ? WT0 : nullptr;
    if (int rc = check(__func__, A)) return rc;
    QT_ARG(W1 || W0, "no output requested");
    const int K2 = 2 * K - 1, ksp = (K + 3) / 4 * 4, nc = 4 * h;
    const int64_t n = (W1 ? (int64_t)(K2 * (cin_pad + h) + ksp) * nc : 0) + (W0 ? (int64_t)(K2 * cin_pad + ksp) * nc : 0);
    hipLaunchKernelGGL(k_compose2_fwd, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_compose2_bwd(const float* Px0, const float* Bx0, const float* Px1, const float* Bx1, const float* Ph0,
                               const float* Bh0, const float* Ph1, const float* Bh1, int K, int cin, int cin_pad, int h,
                               const float* gW1, const float* gW0, float* gPx0, float* gBx0, float* gPx1, float* gBx1,
                               float* gPh0, float* gBh0, float* gPh1, float* gBh1, void* stream) {
    ComposeArgs A = {};
    A.x = {Px0, Bx0, Px1, Bx1};
    A.hb = {Ph0, Bh0, Ph1, Bh1};
    A.K = K; A.cin = cin; A.cin_pad = cin_pad; A.h = h; A.gW1 = gW1; A.gW0 = gW0;
    A.gxP0 = gPx0; A.gxB0 = gBx0; A.gxP1 = gPx1; A.gxB1 = gBx1;
    A.ghP0 = gPh0; A.ghB0 = gBh0; A.ghP1 = gPh1; A.ghB1 = gBh1;
    if (int rc = check(__func__, A)) return rc;
    QT_ARG(gW1 || gW0, "no gradient given");
    QT_ARG(gPx0 && gBx0 && gPx1 && gBx1 && gPh0 && gBh0 && gPh1 && gBh1, "null gradient pointer");
    const int64_t n = (int64_t)4 * K * cin * h + (int64_t)12 * K * h * h + 16 * h;
    hipLaunchKernelGGL(k_compose2_bwd, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
    QT_LAUNCHED();
    return QT_OK;
}
