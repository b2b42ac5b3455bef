// Device helpers of the peephole graph-LSTM cell shared by lstm.hip (stand-alone cell kernels) and cheb.hip (the gate GEMM
// with the cell fused into its epilogue): model/model.py:394-428 + the LayerNorms of model/seq2seq.py:64-75, 140-151.
#pragma once
#include "qt_common.h"

namespace qtcell {

constexpr float LN_EPS = 1e-5f;

// Branch-free gate nonlinearities on the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each).  The libm forms
// (expf + IEEE division, tanhf with its range branches) cost ~25-40 instructions per value: 80 values per node made the
// cell ~10 us of pure VALU issue per launch at the bench shape, as much as the gate GEMM's MFMA time, and the branches
// kept the scheduler from placing that arithmetic beside the MFMAs.  Error: sigmoid <= 3 ulp; tanh <= 2e-7 absolute
// (<= 8 ulp; |x| < 1/8 takes the odd series, where the quotient form would lose relative accuracy by cancellation).
// Forward and backward call the SAME functions, so recomputed values carry the same bits.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * ax));
    const float small = ax * fmaf(x2, fmaf(x2, fmaf(x2, -17.0f / 315.0f, 2.0f / 15.0f), -1.0f / 3.0f), 1.0f);
    return copysignf(ax < 0.125f ? small : big, x);
}

template <int LPN>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int d = 1; d < LPN; d <<= 1) v += __shfl_xor(v, d, 64);
    return v;
}

struct F4 {
    float v[4];
};
__device__ __forceinline__ F4 ld4(const float* p) {
    const float4 f = *reinterpret_cast<const float4*>(p);
    return F4{{f.x, f.y, f.z, f.w}};
}
__device__ __forceinline__ void st4(float* p, const F4& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}

// y = gamma * xhat + beta over the group's h values; returns xhat and rstd
template <int LPN>
__device__ __forceinline__ void layer_norm(const F4& x, int h, F4* xhat, float* rstd) {
    const float mean = group_sum<LPN>((x.v[0] + x.v[1]) + (x.v[2] + x.v[3])) / (float)h;
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float d = x.v[k] - mean;
        sq += d * d;
    }
    const float var = group_sum<LPN>(sq) / (float)h;
    *rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
    for (int k = 0; k < 4; ++k) xhat->v[k] = (x.v[k] - mean) * (*rstd);
}

// gx = rstd * (gxh - mean(gxh) - xhat * mean(gxh * xhat)),  gxh = gy * gamma
template <int LPN>
__device__ __forceinline__ F4 layer_norm_bwd(const F4& gy, const F4& gamma, const F4& xhat, float rstd, int h) {
    F4 gxh;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        gxh.v[k] = gy.v[k] * gamma.v[k];
        s1 += gxh.v[k];
        s2 += gxh.v[k] * xhat.v[k];
    }
    s1 = group_sum<LPN>(s1) / (float)h;
    s2 = group_sum<LPN>(s2) / (float)h;
    F4 gx;
#pragma unroll
    for (int k = 0; k < 4; ++k) gx.v[k] = rstd * (gxh.v[k] - s1 - xhat.v[k] * s2);
    return gx;
}

// C' before its LayerNorm.  The backward recomputes it from the saved gate activations instead of reading a saved copy, so
// the rounding is pinned here (one fma, not left to the compiler's contraction choice): both sides get the same bits.
__device__ __forceinline__ float cell_craw(float F, float cp, float I, float T) { return fmaf(F, cp, I * T); }

// One cell update for 4 hidden units of a node (this lane's slice j0 .. j0+3 of the h channels; the node's h/4 lanes are
// adjacent, LayerNorm statistics are shuffles inside that group -- every lane of the group must call this).
struct CellOut {
    F4 I, F, T, Og, Cr, hn, cn;
};
// The cell's parameters for this lane's 4 hidden units, loaded once (a persistent kernel keeps them in registers: a load in
// the epilogue would make the wave wait for every older load in its queue, the prefetched operand stream included).
struct CellParams {
    F4 wci, wcf, wco, bi, bf, bc, bo, gh, bh, gcn, bcn;
    bool has_ln;
};
__device__ __forceinline__ CellParams cell_params(const float* __restrict__ wc, const float* __restrict__ b,
                                                  const float* __restrict__ ln, int h, int j0) {
    CellParams p;
    p.wci = ld4(wc + j0); p.wcf = ld4(wc + h + j0); p.wco = ld4(wc + 2 * h + j0);
    p.bi = ld4(b + j0); p.bf = ld4(b + h + j0); p.bc = ld4(b + 2 * h + j0); p.bo = ld4(b + 3 * h + j0);
    p.has_ln = ln != nullptr;
    const F4 one = {{1, 1, 1, 1}}, zero = {{0, 0, 0, 0}};
    p.gh = p.gcn = one;
    p.bh = p.bcn = zero;
    if (ln) {
        p.gh = ld4(ln + j0); p.bh = ld4(ln + h + j0); p.gcn = ld4(ln + 2 * h + j0); p.bcn = ld4(ln + 3 * h + j0);
    }
    return p;
}

template <int LPN>
__device__ __forceinline__ CellOut cell_forward(const F4& gi, const F4& gf, const F4& gc, const F4& go, const F4& cp,
                                                const CellParams& P, int h) {
    CellOut r;
    F4 Hr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r.I.v[k] = sigmoidf_(gi.v[k] + P.wci.v[k] * cp.v[k] + P.bi.v[k]);
        r.F.v[k] = sigmoidf_(gf.v[k] + P.wcf.v[k] * cp.v[k] + P.bf.v[k]);
        r.T.v[k] = tanhf_(gc.v[k] + P.bc.v[k]);
        r.Cr.v[k] = cell_craw(r.F.v[k], cp.v[k], r.I.v[k], r.T.v[k]);
        r.Og.v[k] = sigmoidf_(go.v[k] + P.wco.v[k] * r.Cr.v[k] + P.bo.v[k]);
        Hr.v[k] = r.Og.v[k] * tanhf_(r.Cr.v[k]);
    }
    r.hn = Hr;
    r.cn = r.Cr;
    if (P.has_ln) {
        F4 xh, xc;
        float rh, rc;
        layer_norm<LPN>(Hr, h, &xh, &rh);
        layer_norm<LPN>(r.Cr, h, &xc, &rc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r.hn.v[k] = P.gh.v[k] * xh.v[k] + P.bh.v[k];
            r.cn.v[k] = P.gcn.v[k] * xc.v[k] + P.bcn.v[k];
        }
    }
    return r;
}

template <int LPN>
__device__ __forceinline__ CellOut cell_forward(const F4& gi, const F4& gf, const F4& gc, const F4& go, const F4& cp,
                                                const float* __restrict__ wc, const float* __restrict__ b,
                                                const float* __restrict__ ln, int h, int j0) {
    return cell_forward<LPN>(gi, gf, gc, go, cp, cell_params(wc, b, ln, h, j0), h);
}

// Backward of one cell update for 4 hidden units of a node (lane group as in cell_forward).  acc[11][4] collects this
// thread's parameter-gradient terms: w_c i, f, o | b i, f, c, o | LayerNorm gamma_h, beta_h, gamma_c, beta_c.
struct CellBwdOut {
    F4 ggi, ggf, ggc, ggo, gcp;
};
template <int LPN>
__device__ __forceinline__ CellBwdOut cell_backward(const F4& I, const F4& F, const F4& T, const F4& Og, const F4& cp,
                                                    const F4& gyh, const F4& gyc, const F4& go_in, const F4& wci,
                                                    const F4& wcf, const F4& wco, const F4& gam_h, const F4& gam_c,
                                                    bool has_ln, int h, float (&acc)[11][4]) {
    F4 Cr, Hr, tc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        Cr.v[k] = cell_craw(F.v[k], cp.v[k], I.v[k], T.v[k]);        // not saved by the forward: same fma, same bits
        tc.v[k] = tanhf_(Cr.v[k]);
        Hr.v[k] = Og.v[k] * tc.v[k];
    }
    F4 xh = {{0, 0, 0, 0}}, xc = {{0, 0, 0, 0}};
    F4 gHr = gyh, gCr = gyc;
    if (has_ln) {
        float rh, rc;
        layer_norm<LPN>(Hr, h, &xh, &rh);
        layer_norm<LPN>(Cr, h, &xc, &rc);
        gHr = layer_norm_bwd<LPN>(gyh, gam_h, xh, rh, h);
        gCr = layer_norm_bwd<LPN>(gyc, gam_c, xc, rc, h);
    }
    CellBwdOut o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc[7][k] += gyh.v[k] * xh.v[k];
        acc[8][k] += gyh.v[k];
        acc[9][k] += gyc.v[k] * xc.v[k];
        acc[10][k] += gyc.v[k];
        const float gOt = go_in.v[k] + gHr.v[k] * tc.v[k];
        float gc_ = gCr.v[k] + gHr.v[k] * Og.v[k] * (1.0f - tc.v[k] * tc.v[k]);
        o.ggo.v[k] = gOt * Og.v[k] * (1.0f - Og.v[k]);
        gc_ += o.ggo.v[k] * wco.v[k];
        o.ggi.v[k] = gc_ * T.v[k] * I.v[k] * (1.0f - I.v[k]);
        o.ggf.v[k] = gc_ * cp.v[k] * F.v[k] * (1.0f - F.v[k]);
        o.ggc.v[k] = gc_ * I.v[k] * (1.0f - T.v[k] * T.v[k]);
        o.gcp.v[k] = gc_ * F.v[k] + o.ggi.v[k] * wci.v[k] + o.ggf.v[k] * wcf.v[k];
        acc[0][k] += o.ggi.v[k] * cp.v[k];
        acc[1][k] += o.ggf.v[k] * cp.v[k];
        acc[2][k] += o.ggo.v[k] * Cr.v[k];
        acc[3][k] += o.ggi.v[k];
        acc[4][k] += o.ggf.v[k];
        acc[5][k] += o.ggc.v[k];
        acc[6][k] += o.ggo.v[k];
    }
    return o;
}

// reduce NACC*4 per-thread accumulators over a 256-thread workgroup into part_row[NACC * h]; sm: 4 * LPN * NACC * 4 floats
template <int LPN, int NACC>
__device__ __forceinline__ void block_param_reduce(float (&acc)[NACC][4], int h, float* sm, float* part_row, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = acc[a][k];
#pragma unroll
            for (int d = LPN; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
            acc[a][k] = v;
        }
    if (lane < LPN) {
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) sm[(wave * LPN + lane) * NACC * 4 + a * 4 + k] = acc[a][k];
    }
    qt_lds_barrier();                      // (only `sm` crosses: the rows' global stores issued before stay in flight)
    for (int idx = threadIdx.x; idx < NACC * h; idx += 256) {
        const int a = idx / h, j = idx % h;
        const int li = j >> 2, k = j & 3;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += sm[(w * LPN + li) * NACC * 4 + a * 4 + k];
        part_row[idx] = accumulate ? part_row[idx] + s : s;
    }
}

}  // namespace qtcell
