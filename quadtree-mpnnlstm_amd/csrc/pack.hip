// Gate-weight packing of a GConvLSTM whose eight GraphConv stacks hold SEVERAL ChebConvs each (model/model.py:59-97, no
// nonlinearity in between, :95-96): the stacks are composed in weight space into one Chebyshev series,
//   T_a T_b = (T_{a+b} + T_|a-b|) / 2   =>   M[k] = sum_{a,b} cf(k,a,b) P[a] W[b],
// layer by layer.  k_compose_step does one such product for one branch in the natural layout (series (4, Ka, in, h) times
// layer (4, K, h, h) -> series (4, Ka + K - 1, in, h), with the bias series riding along); k_compose2 does the LAST product
// of both branches and writes it straight into the packed gate matrix W (K' C + pad4(bias orders), 4h) that k_gemm_fwd
// multiplies [T_k(L^) Z | T_k(L^) 1] with (Z = [X | H], columns gate-major i, f, c, o).  Two layers per stack (the
// reference's default) are one k_compose2 launch; L layers are L - 2 steps per branch and one k_compose2.
//
// Parameter-sized work (a few thousand outputs of a few hundred FMAs): one launch forward, one backward per product,
// instead of the ~110 bmm / pad / cat / slice kernels the same algebra costs through torch ops and their autograd.
#include "qt_common.h"

namespace {

struct Branch {
    const float* P0;   // (4, Ka, in, h)   series so far (first layer: lins^T per gate, Ka = K)
    const float* B0;   // (4, Kb0, h)      its bias series (first layer: Kb0 = 1)
    const float* P1;   // (4, K, h, h)     next layer
    const float* B1;   // (4, h)
};

struct ComposeArgs {
    Branch x, hb;
    int Ka, Kb0, K, cin, cin_pad, h;
    float* W1;         // variant with H: (K2 (cin_pad + h) + ksp, 4h), or NULL
    float* W0;         // variant without H: (K2 cin_pad + ksp, 4h), or NULL
    float* WT1;        // the transposes (4h, rows), optional: what the gate GEMM stages its weight chunk from
    float* WT0;
    // backward
    const float* gW1;
    const float* gW0;
    float *gxP0, *gxB0, *gxP1, *gxB1, *ghP0, *ghB0, *ghP1, *ghB1;
};

struct StepArgs {      // one branch, natural layout
    Branch b;
    int Ka, Kb0, K, in, h;
    float* P;          // (4, Ka + K - 1, in, h)
    float* B;          // (4, Kb0 + K - 1, h)
    const float* gP;
    const float* gB;
    float *gP0, *gB0, *gP1, *gB1;
};

__device__ __forceinline__ float cf(int k, int a, int b) {
    return 0.5f * (float)((a + b == k) + ((a > b ? a - b : b - a) == k));
}

// M[g, k, c, p] of one branch
__device__ float series_elem(const Branch& br, int g, int k, int c, int p, int in, int Ka, int K, int h) {
    float s = 0.0f;
    for (int a = 0; a < Ka; ++a)
        for (int b = 0; b < K; ++b) {
            const float f = cf(k, a, b);
            if (f == 0.0f) continue;
            const float* r0 = br.P0 + ((int64_t)(g * Ka + a) * in + c) * h;
            const float* r1 = br.P1 + (int64_t)(g * K + b) * h * h + p;
            float d = 0.0f;
            for (int o = 0; o < h; ++o) d = fmaf(r0[o], r1[(int64_t)o * h], d);
            s = fmaf(f, d, s);
        }
    return s;
}

// bias series order k (< Kb0 + K - 1) of one branch: the bias series so far through the layer, plus its bias at order 0
__device__ float bias_elem(const Branch& br, int g, int k, int p, int Kb0, int K, int h) {
    float s = 0.0f;
    for (int a = 0; a < Kb0; ++a)
        for (int b = 0; b < K; ++b) {
            const float f = cf(k, a, b);
            if (f == 0.0f) continue;
            const float* r0 = br.B0 + (int64_t)(g * Kb0 + a) * h;
            const float* r1 = br.P1 + (int64_t)(g * K + b) * h * h + p;
            float d = 0.0f;
            for (int o = 0; o < h; ++o) d = fmaf(r0[o], r1[(int64_t)o * h], d);
            s = fmaf(f, d, s);
        }
    return k == 0 ? s + br.B1[g * h + p] : s;
}

__global__ __launch_bounds__(256) void k_compose2_fwd(ComposeArgs A) {
    const int K = A.K, K2 = A.Ka + K - 1, Kbo = A.Kb0 + K - 1, h = A.h, nc = 4 * h, ksp = (Kbo + 3) / 4 * 4;
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
            v = series_elem(A.x, g, k, c, p, A.cin, A.Ka, K, h);
        else if (c >= A.cin_pad)
            v = series_elem(A.hb, g, k, c - A.cin_pad, p, h, A.Ka, K, h);
    } else {
        const int k = row - K2 * C;
        if (k < Kbo) v = bias_elem(A.x, g, k, p, A.Kb0, K, h) + bias_elem(A.hb, g, k, p, A.Kb0, K, h);
    }
    W[idx] = v;
    if (WT) WT[(int64_t)col * rows + row] = v;
}

__global__ __launch_bounds__(256) void k_compose_step_fwd(StepArgs A) {
    const int K2 = A.Ka + A.K - 1, Kbo = A.Kb0 + A.K - 1, h = A.h;
    const int nP = 4 * K2 * A.in * h, nB = 4 * Kbo * h;
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < nP) {
        const int p = idx % h, c = (idx / h) % A.in, k = (idx / (h * A.in)) % K2, g = idx / (h * A.in * K2);
        A.P[idx] = series_elem(A.b, g, k, c, p, A.in, A.Ka, A.K, h);
        return;
    }
    idx -= nP;
    if (idx < nB) {
        const int p = idx % h, k = (idx / h) % Kbo, g = idx / (h * Kbo);
        A.B[idx] = bias_elem(A.b, g, k, p, A.Kb0, A.K, h);
    }
}

// dL/dM[g, k, c, p] and dL/dbias-series[k, g, p] as the two output layouts hold them
struct PackedGrad {            // packed gate matrices (both variants summed)
    const float *gW1, *gW0;
    int cin_pad, h, K2;
    bool hb;
    __device__ __forceinline__ float gm(int k, int c, int g, int p) const {
        const int nc = 4 * h, col = g * h + p;
        if (hb) return gW1 ? gW1[(int64_t)(k * (cin_pad + h) + cin_pad + c) * nc + col] : 0.0f;
        float v = 0.0f;
        if (gW1) v += gW1[(int64_t)(k * (cin_pad + h) + c) * nc + col];
        if (gW0) v += gW0[(int64_t)(k * cin_pad + c) * nc + col];
        return v;
    }
    __device__ __forceinline__ float gb(int k, int g, int p) const {
        const int nc = 4 * h, col = g * h + p;
        float v = 0.0f;
        if (gW1) v += gW1[(int64_t)(K2 * (cin_pad + h) + k) * nc + col];
        if (gW0) v += gW0[(int64_t)(K2 * cin_pad + k) * nc + col];
        return v;
    }
};
struct NaturalGrad {           // (4, K2, in, h) and (4, Kbo, h)
    const float *gP, *gB;
    int in, h, K2, Kbo;
    __device__ __forceinline__ float gm(int k, int c, int g, int p) const { return gP[((int64_t)(g * K2 + k) * in + c) * h + p]; }
    __device__ __forceinline__ float gb(int k, int g, int p) const { return gB[(int64_t)(g * Kbo + k) * h + p]; }
};

// gradient element `idx` of one branch's four inputs, laid out [gP0 | gP1 | gB0 | gB1]
template <class G>
__device__ void branch_bwd(const Branch& br, const G& gr, int Ka, int Kb0, int K, int in, int h, float* gP0, float* gB0,
                           float* gP1, float* gB1, int idx) {
    const int K2 = Ka + K - 1, Kbo = Kb0 + K - 1;
    const int nP0 = 4 * Ka * in * h, nP1 = 4 * K * h * h, nB0 = 4 * Kb0 * h, nB1 = 4 * h;
    if (idx < nP0) {                         // gP0[g, a, c, o] = sum_{b, k} cf sum_p gM[g, k, c, p] P1[g, b, o, p]
        const int o = idx % h, c = (idx / h) % in, a = (idx / (h * in)) % Ka, g = idx / (h * in * Ka);
        float s = 0.0f;
        for (int b = 0; b < K; ++b)
            for (int k = 0; k < K2; ++k) {
                const float f = cf(k, a, b);
                if (f == 0.0f) continue;
                const float* r1 = br.P1 + ((int64_t)(g * K + b) * h + o) * h;
                float d = 0.0f;
                for (int p = 0; p < h; ++p) d = fmaf(gr.gm(k, c, g, p), r1[p], d);
                s = fmaf(f, d, s);
            }
        gP0[idx] = s;
        return;
    }
    idx -= nP0;
    if (idx < nP1) {           // gP1[g, b, o, p] = sum_{a, k} cf (sum_c P0[g, a, c, o] gM[g, k, c, p]) + sum_{a, k} cf B0[g, a, o] gBias[k, g, p]
        const int p = idx % h, o = (idx / h) % h, b = (idx / (h * h)) % K, g = idx / (h * h * K);
        float s = 0.0f;
        for (int a = 0; a < Ka; ++a)
            for (int k = 0; k < K2; ++k) {
                const float f = cf(k, a, b);
                if (f == 0.0f) continue;
                const float* r0 = br.P0 + (int64_t)(g * Ka + a) * in * h + o;
                float d = 0.0f;
                for (int c = 0; c < in; ++c) d = fmaf(r0[(int64_t)c * h], gr.gm(k, c, g, p), d);
                s = fmaf(f, d, s);
            }
        for (int a = 0; a < Kb0; ++a)
            for (int k = 0; k < Kbo; ++k) {
                const float f = cf(k, a, b);
                if (f != 0.0f) s = fmaf(f * br.B0[(int64_t)(g * Kb0 + a) * h + o], gr.gb(k, g, p), s);
            }
        gP1[idx] = s;
        return;
    }
    idx -= nP1;
    if (idx < nB0) {                         // gB0[g, a, o] = sum_{b, k} cf sum_p gBias[k, g, p] P1[g, b, o, p]
        const int o = idx % h, a = (idx / h) % Kb0, g = idx / (h * Kb0);
        float s = 0.0f;
        for (int b = 0; b < K; ++b)
            for (int k = 0; k < Kbo; ++k) {
                const float f = cf(k, a, b);
                if (f == 0.0f) continue;
                const float* r1 = br.P1 + ((int64_t)(g * K + b) * h + o) * h;
                float d = 0.0f;
                for (int p = 0; p < h; ++p) d = fmaf(gr.gb(k, g, p), r1[p], d);
                s = fmaf(f, d, s);
            }
        gB0[idx] = s;
        return;
    }
    idx -= nB0;
    if (idx < nB1) gB1[idx] = gr.gb(0, idx / h, idx % h);
}

__device__ __forceinline__ int branch_count(int Ka, int Kb0, int K, int in, int h) {
    return 4 * Ka * in * h + 4 * K * h * h + 4 * Kb0 * h + 4 * h;
}

__global__ __launch_bounds__(256) void k_compose2_bwd(ComposeArgs A) {
    const int nx = branch_count(A.Ka, A.Kb0, A.K, A.cin, A.h), nh = branch_count(A.Ka, A.Kb0, A.K, A.h, A.h);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    PackedGrad gr = {A.gW1, A.gW0, A.cin_pad, A.h, A.Ka + A.K - 1, false};
    if (idx < nx) {
        branch_bwd(A.x, gr, A.Ka, A.Kb0, A.K, A.cin, A.h, A.gxP0, A.gxB0, A.gxP1, A.gxB1, idx);
    } else if (idx < nx + nh) {
        gr.hb = true;
        branch_bwd(A.hb, gr, A.Ka, A.Kb0, A.K, A.h, A.h, A.ghP0, A.ghB0, A.ghP1, A.ghB1, idx - nx);
    }
}

__global__ __launch_bounds__(256) void k_compose_step_bwd(StepArgs A) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= branch_count(A.Ka, A.Kb0, A.K, A.in, A.h)) return;
    const NaturalGrad gr = {A.gP, A.gB, A.in, A.h, A.Ka + A.K - 1, A.Kb0 + A.K - 1};
    branch_bwd(A.b, gr, A.Ka, A.Kb0, A.K, A.in, A.h, A.gP0, A.gB0, A.gP1, A.gB1, idx);
}

int check_branch(const char* fn, const Branch& b) {
    if (!b.P0 || !b.B0 || !b.P1 || !b.B1) {
        qt_set_error("%s: null weight pointer", fn);
        return QT_E_ARG;
    }
    return QT_OK;
}

int check_sizes(const char* fn, int Ka, int Kb0, int K, int in, int h) {
    if (Ka < 1 || Ka > 32 || Kb0 < 1 || Kb0 > Ka || K < 1 || K > 8 || in < 1 || h < 4 || h % 4) {
        qt_set_error("%s: bad sizes (Ka in 1..32, Kb0 in 1..Ka, K in 1..8, h a multiple of 4)", fn);
        return QT_E_ARG;
    }
    return QT_OK;
}

}  // namespace

extern "C" int qt_compose2_fwd(const float* Px0, const float* Bx0, const float* Px1, const float* Bx1, const float* Ph0,
                               const float* Bh0, const float* Ph1, const float* Bh1, int Ka, int Kb0, int K, int cin,
                               int cin_pad, int h, float* W1, float* W0, float* WT1, float* WT0, void* stream) {
    ComposeArgs A = {};
    A.x = {Px0, Bx0, Px1, Bx1};
    A.hb = {Ph0, Bh0, Ph1, Bh1};
    A.Ka = Ka; A.Kb0 = Kb0; A.K = K; A.cin = cin; A.cin_pad = cin_pad; A.h = h;
    A.W1 = W1; A.W0 = W0; A.WT1 = W1 ? WT1 : nullptr; A.WT0 = W0 ? WT0 : nullptr;
    if (int rc = check_branch(__func__, A.x)) return rc;
    if (int rc = check_branch(__func__, A.hb)) return rc;
    if (int rc = check_sizes(__func__, Ka, Kb0, K, cin, h)) return rc;
    QT_ARG(cin_pad >= cin && cin_pad % 4 == 0, "cin_pad must be a multiple of 4 and >= cin");
    QT_ARG(W1 || W0, "no output requested");
    const int K2 = Ka + K - 1, ksp = (Kb0 + K - 1 + 3) / 4 * 4, nc = 4 * h;
    const int64_t n = (W1 ? (int64_t)(K2 * (cin_pad + h) + ksp) * nc : 0) + (W0 ? (int64_t)(K2 * cin_pad + ksp) * nc : 0);
    hipLaunchKernelGGL(k_compose2_fwd, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_compose2_bwd(const float* Px0, const float* Bx0, const float* Px1, const float* Bx1, const float* Ph0,
                               const float* Bh0, const float* Ph1, const float* Bh1, int Ka, int Kb0, int K, int cin,
                               int cin_pad, int h, const float* gW1, const float* gW0, float* gPx0, float* gBx0,
                               float* gPx1, float* gBx1, float* gPh0, float* gBh0, float* gPh1, float* gBh1, void* stream) {
    ComposeArgs A = {};
    A.x = {Px0, Bx0, Px1, Bx1};
    A.hb = {Ph0, Bh0, Ph1, Bh1};
    A.Ka = Ka; A.Kb0 = Kb0; A.K = K; A.cin = cin; A.cin_pad = cin_pad; A.h = h; A.gW1 = gW1; A.gW0 = gW0;
    A.gxP0 = gPx0; A.gxB0 = gBx0; A.gxP1 = gPx1; A.gxB1 = gBx1;
    A.ghP0 = gPh0; A.ghB0 = gBh0; A.ghP1 = gPh1; A.ghB1 = gBh1;
    if (int rc = check_branch(__func__, A.x)) return rc;
    if (int rc = check_branch(__func__, A.hb)) return rc;
    if (int rc = check_sizes(__func__, Ka, Kb0, K, cin, h)) return rc;
    QT_ARG(cin_pad >= cin && cin_pad % 4 == 0, "cin_pad must be a multiple of 4 and >= cin");
    QT_ARG(gW1 || gW0, "no gradient given");
    QT_ARG(gPx0 && gBx0 && gPx1 && gBx1 && gPh0 && gBh0 && gPh1 && gBh1, "null gradient pointer");
    const int64_t n = (int64_t)4 * Ka * (cin + h) * h + (int64_t)8 * K * h * h + (int64_t)8 * Kb0 * h + 8 * h;
    hipLaunchKernelGGL(k_compose2_bwd, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_compose_step_fwd(const float* P0, const float* B0, const float* P1, const float* B1, int Ka, int Kb0, int K,
                                   int in, int h, float* P, float* B, void* stream) {
    StepArgs A = {};
    A.b = {P0, B0, P1, B1};
    A.Ka = Ka; A.Kb0 = Kb0; A.K = K; A.in = in; A.h = h; A.P = P; A.B = B;
    if (int rc = check_branch(__func__, A.b)) return rc;
    if (int rc = check_sizes(__func__, Ka, Kb0, K, in, h)) return rc;
    QT_ARG(P && B, "null output pointer");
    const int64_t n = (int64_t)4 * (Ka + K - 1) * in * h + (int64_t)4 * (Kb0 + K - 1) * h;
    hipLaunchKernelGGL(k_compose_step_fwd, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_compose_step_bwd(const float* P0, const float* B0, const float* P1, const float* B1, int Ka, int Kb0, int K,
                                   int in, int h, const float* gP, const float* gB, float* gP0, float* gB0, float* gP1,
                                   float* gB1, void* stream) {
    StepArgs A = {};
    A.b = {P0, B0, P1, B1};
    A.Ka = Ka; A.Kb0 = Kb0; A.K = K; A.in = in; A.h = h; A.gP = gP; A.gB = gB;
    A.gP0 = gP0; A.gB0 = gB0; A.gP1 = gP1; A.gB1 = gB1;
    if (int rc = check_branch(__func__, A.b)) return rc;
    if (int rc = check_sizes(__func__, Ka, Kb0, K, in, h)) return rc;
    QT_ARG(gP && gB && gP0 && gB0 && gP1 && gB1, "null gradient pointer");
    const int64_t n = (int64_t)4 * Ka * in * h + (int64_t)4 * K * h * h + (int64_t)4 * Kb0 * h + 4 * h;
    hipLaunchKernelGGL(k_compose_step_bwd, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, A);
    QT_LAUNCHED();
    return QT_OK;
}

// ---- clip_grad_norm_ + Adam on ONE flat parameter vector (model/mpnnlstm.py:251-257 on the flat buffer of qtmpnn/flat.py).
// torch's fused Adam walks a single 34.5k-element tensor with one 512-thread block (52 us per step in the round-2 profile) and
// the clipping is another eight small launches; here: one block sums the squares in a fixed order (bit-reproducible) and bumps
// the step counter, then an elementwise launch scales the gradient by min(1, max_norm / (norm + 1e-6)) -- in place, like
// clip_grad_norm_ -- and applies Adam (no weight decay, no amsgrad: the reference's torch.optim.Adam(lr) defaults).
namespace {

__global__ __launch_bounds__(1024) void k_flat_sumsq(const float* __restrict__ g, int n, float* __restrict__ stat,
                                                     int32_t* __restrict__ step) {
    __shared__ float red[16];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < n; i += 1024) acc += g[i] * g[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += red[w];
        stat[0] = sqrtf(s);                   // total gradient norm (before clipping), what clip_grad_norm_ returns
        *step += 1;
    }
}

__global__ __launch_bounds__(256) void k_flat_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int n, const float* __restrict__ stat,
                                                   const int32_t* __restrict__ step, const float* __restrict__ lr_dev,
                                                   float lr_host, float beta1, float beta2, float eps, float max_norm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float lr = lr_dev ? *lr_dev : lr_host;
    float coef = 1.0f;
    if (max_norm > 0.0f) coef = fminf(max_norm / (stat[0] + 1e-6f), 1.0f);
    const float t = (float)*step;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    g[i] = gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
}

}  // namespace

extern "C" int qt_flat_adam(float* p, float* g, float* m, float* v, int n, int32_t* step, const float* lr_dev, float lr_host,
                            float beta1, float beta2, float eps, float max_norm, float* stat, void* stream) {
    QT_ARG(p && g && m && v && step && stat && n > 0, "bad arguments");
    hipLaunchKernelGGL(k_flat_sumsq, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, n, stat, step);
    QT_LAUNCHED();
    hipLaunchKernelGGL(k_flat_adam, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, stat, step, lr_dev,
                       lr_host, beta1, beta2, eps, max_norm);
    QT_LAUNCHED();
    return QT_OK;
}

// ---- x = hi + lo with two bf16 terms (hi = round(x), lo = round(x - hi)): the right operand of a split-bf16 MFMA product
// (qt_lstm_bwd_dgrad's data gradient), split once per pass instead of per workgroup.
namespace {
__global__ __launch_bounds__(256) void k_split_bf16(const float* __restrict__ x, int64_t n, __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const __bf16 h = (__bf16)v;
    hi[i] = h;
    lo[i] = (__bf16)(v - (float)h);
}
}  // namespace

extern "C" int qt_split_bf16(const float* x, int64_t n, void* hi, void* lo, void* stream) {
    QT_ARG(x && hi && lo && n > 0, "bad arguments");
    hipLaunchKernelGGL(k_split_bf16, dim3(qt_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, n, (__bf16*)hi, (__bf16*)lo);
    QT_LAUNCHED();
    return QT_OK;
}
