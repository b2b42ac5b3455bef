// Peephole graph-LSTM cell (model/model.py:394-428) fused with the LayerNorms the
// encoder / decoder apply to H' and C' (model/seq2seq.py:64-75, 140-151), and the decoder
// head's norm_o + relu + concat (model/seq2seq.py:160-165).
//
// A node's h channels are spread over h/4 adjacent lanes (float4 each); LayerNorm
// statistics are xor-shuffle reductions inside that lane group.  Parameter gradients are
// accumulated per thread, reduced over the wave with shuffles and over the workgroup
// through LDS in a fixed order, and leave as one partial row per workgroup.
#include "qt_cell.h"

namespace {
using namespace qtcell;

template <int LPN>
__global__ __launch_bounds__(256) void k_lstm_fwd(const float* __restrict__ G, const float* __restrict__ G2, int ld_g,
                                                  const float* __restrict__ Cprev,
                                                  const float* __restrict__ wc, const float* __restrict__ b,
                                                  const float* __restrict__ ln, int Ncap, const int32_t* __restrict__ n_dev,
                                                  int h, int ld_c, float* __restrict__ O, float* __restrict__ Hn,
                                                  float* __restrict__ Cn, float* __restrict__ gates) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t node = gid / LPN;
    if (node >= qt_rows(n_dev, Ncap)) return;
    const int j0 = (int)(gid % LPN) * 4;
    const float* g = G + node * ld_g + j0;
    F4 gi = ld4(g), gf = ld4(g + h), gc = ld4(g + 2 * h), go = ld4(g + 3 * h);
    if (G2) {                   // conv_x(X) + conv_h(H) of the four gates (model/model.py:394-424), summed here
        const float* g2 = G2 + node * ld_g + j0;
        const F4 hi = ld4(g2), hf = ld4(g2 + h), hc = ld4(g2 + 2 * h), ho = ld4(g2 + 3 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            gi.v[c] += hi.v[c];
            gf.v[c] += hf.v[c];
            gc.v[c] += hc.v[c];
            go.v[c] += ho.v[c];
        }
    }
    F4 cp = {{0, 0, 0, 0}};
    if (Cprev) cp = ld4(Cprev + node * ld_c + j0);
    const CellOut r = cell_forward<LPN>(gi, gf, gc, go, cp, wc, b, ln, h, j0);
    const F4 &I = r.I, &F = r.F, &T = r.T, &Og = r.Og, &hn = r.hn, &cn = r.cn;
    st4(O + node * h + j0, Og);
    st4(Hn + node * h + j0, hn);
    st4(Cn + node * h + j0, cn);
    float* gs = gates + node * 4 * h + j0;
    st4(gs, I);
    st4(gs + h, F);
    st4(gs + 2 * h, T);
    st4(gs + 3 * h, Og);
}

template <int LPN>
__global__ __launch_bounds__(256) void k_lstm_bwd(const float* __restrict__ gO, const float* __restrict__ gHn,
                                                  const float* __restrict__ gCn, const float* __restrict__ gates,
                                                  const float* __restrict__ Cprev,
                                                  const float* __restrict__ wc, const float* __restrict__ ln, int Ncap,
                                                  const int32_t* __restrict__ n_dev, int h, int ld_go, int ld_gh, int ld_gc,
                                                  int ld_c, float* __restrict__ gG, float* __restrict__ gCprev,
                                                  float* __restrict__ part, int accumulate) {
    __shared__ float sm[4 * LPN * 11 * 4];
    const int N = qt_rows(n_dev, Ncap);
    const int j0 = (threadIdx.x % LPN) * 4;
    const F4 wci = ld4(wc + j0), wcf = ld4(wc + h + j0), wco = ld4(wc + 2 * h + j0);
    F4 gam_h = {{1, 1, 1, 1}}, gam_c = {{1, 1, 1, 1}};
    if (ln) {
        gam_h = ld4(ln + j0);
        gam_c = ld4(ln + 2 * h + j0);
    }
    float acc[11][4];
#pragma unroll
    for (int a = 0; a < 11; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[a][k] = 0.0f;

    const int64_t stride = (int64_t)gridDim.x * (256 / LPN);
    for (int64_t node = (int64_t)blockIdx.x * (256 / LPN) + threadIdx.x / LPN; node < N; node += stride) {
        const float* gs = gates + node * 4 * h + j0;
        const F4 I = ld4(gs), F = ld4(gs + h), T = ld4(gs + 2 * h), Og = ld4(gs + 3 * h);
        F4 cp = {{0, 0, 0, 0}};
        if (Cprev) cp = ld4(Cprev + node * ld_c + j0);
        F4 gyh = {{0, 0, 0, 0}}, gyc = {{0, 0, 0, 0}};      // an output nobody used has no gradient
        if (gHn) gyh = ld4(gHn + node * ld_gh + j0);
        if (gCn) gyc = ld4(gCn + node * ld_gc + j0);
        F4 go_in = {{0, 0, 0, 0}};
        if (gO) go_in = ld4(gO + node * ld_go + j0);
        const CellBwdOut r = cell_backward<LPN>(I, F, T, Og, cp, gyh, gyc, go_in, wci, wcf, wco, gam_h, gam_c, ln != nullptr, h, acc);
        const F4 &ggi = r.ggi, &ggf = r.ggf, &ggc = r.ggc, &ggo = r.ggo, &gcp = r.gcp;
        float* gg = gG + node * 4 * h + j0;
        st4(gg, ggi);
        st4(gg + h, ggf);
        st4(gg + 2 * h, ggc);
        st4(gg + 3 * h, ggo);
        if (gCprev) st4(gCprev + node * h + j0, gcp);
    }
    block_param_reduce<LPN, 11>(acc, h, sm, part + (int64_t)blockIdx.x * 11 * h, accumulate);
}

template <int LPN>
__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ O, const float* __restrict__ ln_o,
                                                  const float* __restrict__ concat, int Ncap,
                                                  const int32_t* __restrict__ n_dev, int h, int ld_o, int hp,
                                                  float* __restrict__ Z, float* __restrict__ Zb) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t node = gid / LPN;
    if (node >= qt_rows(n_dev, Ncap)) return;
    const int li = (int)(gid % LPN), j0 = li * 4;
    const F4 x = ld4(O + node * ld_o + j0);
    F4 xh;
    float r;
    layer_norm<LPN>(x, h, &xh, &r);
    const F4 gm = ld4(ln_o + j0), bt = ld4(ln_o + h + j0);
    F4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) y.v[k] = fmaxf(gm.v[k] * xh.v[k] + bt.v[k], 0.0f);
    if (Zb) {               // two matrices: (N, h) and (N, hp - h)
        st4(Z + node * h + j0, y);
        if (li == 0)
            for (int j = h; j < hp; ++j) Zb[node * (hp - h) + (j - h)] = (j == h && concat) ? concat[node] : 0.0f;
        return;
    }
    st4(Z + node * hp + j0, y);
    if (li == 0)
        for (int j = h; j < hp; ++j) Z[node * hp + j] = (j == h && concat) ? concat[node] : 0.0f;
}

template <int LPN>
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ gZ, const float* __restrict__ gZb,
                                                  const float* __restrict__ O, const float* __restrict__ ln_o, int Ncap,
                                                  const int32_t* __restrict__ n_dev, int h, int ld_o, int hp,
                                                  float* __restrict__ gO,
                                                  float* __restrict__ gconcat, float* __restrict__ part, int accumulate) {
    __shared__ float sm[4 * LPN * 2 * 4];
    const int N = qt_rows(n_dev, Ncap);
    const int li = threadIdx.x % LPN, j0 = li * 4;
    const F4 gm = ld4(ln_o + j0), bt = ld4(ln_o + h + j0);
    float acc[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[0][k] = acc[1][k] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * (256 / LPN);
    for (int64_t node = (int64_t)blockIdx.x * (256 / LPN) + threadIdx.x / LPN; node < N; node += stride) {
        const F4 x = ld4(O + node * ld_o + j0);
        F4 xh;
        float r;
        layer_norm<LPN>(x, h, &xh, &r);
        F4 gy = ld4(gZ + node * (gZb ? h : hp) + j0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (gm.v[k] * xh.v[k] + bt.v[k] <= 0.0f) gy.v[k] = 0.0f;
            acc[0][k] += gy.v[k] * xh.v[k];
            acc[1][k] += gy.v[k];
        }
        st4(gO + node * h + j0, layer_norm_bwd<LPN>(gy, gm, xh, r, h));
        if (li == 0 && gconcat) gconcat[node] = gZb ? gZb[node * (hp - h)] : gZ[node * hp + h];
    }
    block_param_reduce<LPN, 2>(acc, h, sm, part + (int64_t)blockIdx.x * 2 * h, accumulate);
}

// G = gY * act'(Y) for the GEMM epilogue activations (QT_ACT_RELU / QT_ACT_TANH_RES); for TANH_RES also the gradient of
// the residual operand, gres (N, rs): column 0 = gY[:, 0], the rest 0.  One launch instead of ~8 elementwise tensor ops.
__global__ void k_act_bwd(const float* __restrict__ gY, const float* __restrict__ Y, const float* __restrict__ res, int rs,
                          const float* __restrict__ drop, int act, int Ncap, const int32_t* __restrict__ n_dev, int Co,
                          float* __restrict__ G, float* __restrict__ gres, const float* __restrict__ gY2) {
    // 32-bit thread index and division (the launcher checks N * Co < 2^31): a 64-bit division by a run-time value is a
    // ~100-instruction routine per thread
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned row = idx / (unsigned)Co;
    if ((int)row >= qt_rows(n_dev, Ncap)) return;
    const int c = (int)(idx - row * (unsigned)Co);
    const float g = gY2 ? gY[idx] + gY2[idx] : gY[idx], y = Y[idx];      // (gY2: the gradient of Y's second consumer, same layout)
    float o;
    if (act == QT_ACT_RELU) {
        o = y > 0.0f ? g : 0.0f;
    } else {
        const float t = y - res[row * rs];
        o = g * (1.0f - t * t) * (drop ? drop[row] : 1.0f);
        if (gres && c < rs) gres[row * rs + c] = c == 0 ? g : 0.0f;
    }
    G[idx] = o;
}

// Column concatenation of up to 8 row-strided sources (widths multiples of 4) into one dense (N, C) matrix: Z = [X | H],
// the re-mesh state [out | H_0 .. | C_0 ..].  torch.cat's batched copy moves these narrow rows at ~0.6 TB/s.
struct ConcatArgs {
    const float* src[8];
    int ld[8];
    int c4_end[8];      // exclusive prefix of the widths, in float4 units
    int nsrc, C4;
};
__global__ void k_concat(ConcatArgs a, int Ncap, const int32_t* __restrict__ n_dev, float* __restrict__ out) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;          // (N * C4 < 2^31 checked by the launcher)
    const int64_t row = idx / (unsigned)a.C4;
    if (row >= qt_rows(n_dev, Ncap)) return;
    const int q = (int)(idx - (unsigned)row * (unsigned)a.C4);
    int s = 0;
    while (q >= a.c4_end[s]) ++s;
    const int q0 = s ? a.c4_end[s - 1] : 0;
    reinterpret_cast<float4*>(out)[idx] = *reinterpret_cast<const float4*>(a.src[s] + row * a.ld[s] + 4 * (q - q0));
}

// out[i] = (val4[i].x, pos[i]) -- or (val4[i].x, 0, 0, 0) when pos == NULL (the backward of the same op)
__global__ void k_decoder_input(const float* __restrict__ val4, int ld, const float* __restrict__ pos, int Ncap,
                                const int32_t* __restrict__ n_dev, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qt_rows(n_dev, Ncap)) return;
    float4 o = make_float4(val4[ld * (int64_t)i], 0.0f, 0.0f, 0.0f);
    if (pos) {
        o.y = pos[3 * (int64_t)i];
        o.z = pos[3 * (int64_t)i + 1];
        o.w = pos[3 * (int64_t)i + 2];
    }
    reinterpret_cast<float4*>(out)[i] = o;
}

inline int lanes_per_node(int h) { return h / 4; }
inline bool h_ok(int h) { return h == 8 || h == 16 || h == 32 || h == 64 || h == 128; }

}  // namespace

#define QT_DISPATCH_LPN(h, KERNEL, grid, stream, ...)                                                          \
    switch (lanes_per_node(h)) {                                                                               \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;   \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;   \
        case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;   \
        case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
        default: hipLaunchKernelGGL(KERNEL<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break; \
    }

extern "C" int qt_lstm_fwd(const float* G, const float* G2, int ld_g, const float* Cprev, int ld_c, const float* wc, const float* b,
                           const float* ln, int N, const int32_t* n_dev, int h, float* O, float* Hn, float* Cn, float* gates,
                           void* stream) {
    QT_ARG(G && wc && b && O && Hn && Cn && gates, "null pointer");
    if (ld_g == 0) ld_g = 4 * h;
    QT_ARG(ld_g >= 4 * h && ld_g % 4 == 0 && (((uintptr_t)G | (uintptr_t)G2) & 15) == 0, "bad gate row stride / alignment");
    QT_ARG(h_ok(h), "hidden size must be 8, 16, 32, 64 or 128");
    if (N <= 0) return QT_OK;
    const int grid = qt_cdiv((int64_t)N * lanes_per_node(h), 256);
    QT_ARG(!Cprev || (ld_c >= h && ld_c % 4 == 0), "bad Cprev row stride");
    QT_DISPATCH_LPN(h, k_lstm_fwd, grid, stream, G, G2, ld_g, Cprev, wc, b, ln, N, n_dev, h, ld_c, O, Hn, Cn, gates);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_lstm_bwd_blocks(int N, int h) {
    if (N <= 0 || !h_ok(h)) return 0;
    const int need = qt_cdiv((int64_t)N * lanes_per_node(h), 256);
#ifndef QT_LSTM_BLOCKS
#define QT_LSTM_BLOCKS 512
#endif
    return need < QT_LSTM_BLOCKS ? need : QT_LSTM_BLOCKS;
}

extern "C" int qt_lstm_bwd(const float* gO, int ld_go, const float* gHn, int ld_gh, const float* gCn, int ld_gc,
                           const float* gates, const float* Cprev, int ld_c, const float* wc,
                           const float* ln, int N, const int32_t* n_dev, int h, float* gG, float* gCprev, float* part,
                           int accumulate, void* stream) {
    QT_ARG(gates && wc && gG && part, "null pointer");
    QT_ARG(h_ok(h), "hidden size must be 8, 16, 32, 64 or 128");
    if (N <= 0) return QT_OK;
    const int grid = qt_lstm_bwd_blocks(N, h);
    QT_ARG((!gHn || ld_gh >= h) && (!gCn || ld_gc >= h) && ld_gh % 4 == 0 && ld_gc % 4 == 0 && ld_go % 4 == 0 && ld_c % 4 == 0, "bad row stride");
    QT_DISPATCH_LPN(h, k_lstm_bwd, grid, stream, gO, gHn, gCn, gates, Cprev, wc, ln, N, n_dev, h, ld_go, ld_gh, ld_gc,
                    ld_c, gG, gCprev, part, accumulate);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_head_fwd(const float* O, int ld_o, const float* ln_o, const float* concat, int N, const int32_t* n_dev, int h,
                           int hp, float* Z, float* Zb, void* stream) {
    QT_ARG(O && ln_o && Z, "null pointer");
    QT_ARG(h_ok(h) && hp >= h && hp % 4 == 0 && (!Zb || hp > h), "bad h / hp");
    if (ld_o <= 0) ld_o = h;
    QT_ARG(ld_o % 4 == 0 && ((uintptr_t)O & 15) == 0, "O rows must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    const int grid = qt_cdiv((int64_t)N * lanes_per_node(h), 256);
    QT_DISPATCH_LPN(h, k_head_fwd, grid, stream, O, ln_o, concat, N, n_dev, h, ld_o, hp, Z, Zb);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_head_bwd(const float* gZ, const float* gZb, const float* O, int ld_o, const float* ln_o, int N,
                           const int32_t* n_dev, int h, int hp, float* gO, float* gconcat, float* part, int accumulate,
                           void* stream) {
    QT_ARG(gZ && O && ln_o && gO && part, "null pointer");
    QT_ARG(h_ok(h) && hp >= h && hp % 4 == 0 && (!gZb || hp > h), "bad h / hp");
    if (ld_o <= 0) ld_o = h;
    QT_ARG(ld_o % 4 == 0 && ((uintptr_t)O & 15) == 0, "O rows must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    const int grid = qt_lstm_bwd_blocks(N, h);
    QT_DISPATCH_LPN(h, k_head_bwd, grid, stream, gZ, gZb, O, ln_o, N, n_dev, h, ld_o, hp, gO, gconcat, part, accumulate);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_act_bwd(const float* gY, const float* Y, const float* res, int res_stride, const float* drop, int act, int N,
                          const int32_t* n_dev, int Co, float* G, float* gres, const float* gY2, void* stream) {
    QT_ARG(gY && Y && G && Co >= 1, "null pointer");
    QT_ARG(act == QT_ACT_RELU || (act == QT_ACT_TANH_RES && res && res_stride >= 1 && res_stride <= Co), "bad activation arguments");
    QT_ARG((int64_t)N * Co + 256 < ((int64_t)1 << 31), "N * Co too large for 32-bit thread indices");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_act_bwd, dim3(qt_cdiv((int64_t)N * Co, 256)), dim3(256), 0, (hipStream_t)stream, gY, Y, res, res_stride,
                       drop, act, N, n_dev, Co, G, gres, gY2);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_concat(const float* const* srcs, const int* widths, const int* lds, int nsrc, int N, const int32_t* n_dev,
                         float* out, void* stream) {
    QT_ARG(srcs && widths && lds && out && nsrc >= 1 && nsrc <= 8, "1..8 sources");
    ConcatArgs a;
    int c4 = 0;
    for (int i = 0; i < nsrc; ++i) {
        QT_ARG(srcs[i] && widths[i] > 0 && widths[i] % 4 == 0 && lds[i] % 4 == 0 && lds[i] >= widths[i] && ((uintptr_t)srcs[i] & 15) == 0,
               "sources must be 16-byte aligned with widths / row strides that are multiples of 4");
        a.src[i] = srcs[i];
        a.ld[i] = lds[i];
        c4 += widths[i] / 4;
        a.c4_end[i] = c4;
    }
    for (int i = nsrc; i < 8; ++i) { a.src[i] = nullptr; a.ld[i] = 0; a.c4_end[i] = c4; }
    a.nsrc = nsrc;
    a.C4 = c4;
    QT_ARG((int64_t)N * c4 + 256 < ((int64_t)1 << 31), "N * C too large for 32-bit thread indices");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_concat, dim3(qt_cdiv((int64_t)N * c4, 256)), dim3(256), 0, (hipStream_t)stream, a, N, n_dev, out);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_decoder_input(const float* val4, int ld, const float* posfeat, int N, const int32_t* n_dev, float* out,
                                void* stream) {
    QT_ARG(val4 && out && ((uintptr_t)out & 15) == 0, "bad arguments");
    if (ld <= 0) ld = 4;
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_decoder_input, dim3(qt_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, val4, ld, posfeat, N, n_dev, out);
    QT_LAUNCHED();
    return QT_OK;
}
