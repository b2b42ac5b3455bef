// Chebyshev graph convolution pieces (PyG ChebConv as used by model/model.py:53,96):
//   k_spmm  -- the CSR message-aggregate  out = alpha * L^ x + beta * p + gamma * q
//   k_gemm_fwd / k_gemm_wgrad -- fp32 MFMA (32x32x2) GEMMs: the gate GEMM Y = [T_0 .. T_{K-1} | S] W and its
//              data gradient; the weight gradient split over row blocks
//   k_colsum -- fixed-order reduction of per-block partial sums
#include "qt_cell.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------ message aggregate
// Thread = (RPT consecutive rows) x (one VEC-wide channel chunk), chunk index fastest (RPT = 1 in every launch: two rows
// per thread, or several 256-thread slices per workgroup, only lengthen the chain of dependent loads rowptr -> col/nrm ->
// x rows that a thread walks -- measured 11.1 vs 10.8 us and 13.3 vs 11.0 us at N = 1.2e5, C = 20).  Measured with PMC
// counters at the same shape: FETCH_SIZE equals the algorithmic reads once workgroups are mapped XCD-wise (below), waves
// spend ~60 % of their cycles waiting on memory and the launch moves ~3 TB/s against the 7 TB/s a plain copy of
// the same buffers reaches.
// One launch serves one or two column parts of the same rows (x, p, q, out of width C each): Z = [X | H] of the
// recurrent cells is propagated as its two matrices, never concatenated.  Part b's workgroups follow part a's.
struct SpmmPart {
    const float* x;
    const float* p;
    const float* q;
    float* out;
    int C, ldx, ldp, ldq, xcd_chunk;      // row strides of x / p / q in floats (out rows are dense)
};
template <int VEC, int RPT, int EPT>
#ifndef QT_SPMM_BS
#define QT_SPMM_BS 64      // one wave per workgroup: 11.08 ms per training step against 11.12 (128) and 11.18 (256)
#endif
__global__ __launch_bounds__(QT_SPMM_BS) void k_spmm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                              const float* __restrict__ nrm, const int4* __restrict__ ell, int Ncap,
                                              const int32_t* __restrict__ n_dev,
                                              SpmmPart pa, SpmmPart pb, int nblk_a,     // workgroups [nblk_a, ..) do part b
                                              float alpha, float beta, float gamma) {
    const bool second = (int)blockIdx.x >= nblk_a;
    const SpmmPart& P = second ? pb : pa;
    const int C = P.C, ldx = P.ldx, xcd_chunk = P.xcd_chunk;
    const float* __restrict__ x = P.x;
    const float* p = P.p;
    const float* q = P.q;
    float* out = P.out;                     // out may alias p or q (in-place Clenshaw step)
    const int bid = second ? (int)blockIdx.x - nblk_a : (int)blockIdx.x;
    const int nch = C / VEC;
    // Workgroups are dealt round-robin to the 8 XCDs, each with a private L2.  Giving XCD x the contiguous node range
    // [x * chunk, (x+1) * chunk) keeps a node's neighbours (close in the reversed-Morton order) in the L2 that reads them.
    // The chunk is an eighth of the VALID rows (read on the device in static mode), not of the capacity the grid was
    // sized for: otherwise the last XCDs idle whenever a mesh has fewer nodes than pixels.
    const int rows = qt_rows(n_dev, Ncap);
    int blk = bid;
    if (xcd_chunk) {
        const int nblk = (int)(((int64_t)((rows + RPT - 1) / RPT) * nch + QT_SPMM_BS - 1) / QT_SPMM_BS);
        const int chunk = (nblk + 7) >> 3;
        if ((bid >> 3) >= chunk) return;
        blk = (bid & 7) * chunk + (bid >> 3);
    }
    const unsigned idx = (unsigned)blk * (unsigned)QT_SPMM_BS + threadIdx.x;      // N * nch < 2^31 (checked by the host entry): 32-bit
    const int64_t rp = idx / (unsigned)nch;                        // division, a fraction of the 64-bit one's cost
    if (rp * RPT >= rows) return;
    const int ch = (int)(idx - (unsigned)rp * (unsigned)nch) * VEC;
    int e0[RPT], e1[RPT];
    int64_t row[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        row[u] = rp * RPT + u;
        const bool ok = row[u] < rows;
        if (!ok) row[u] = rows - 1;                // duplicate of a valid row; its result is not stored
        e0[u] = (ok && !(VEC == 4 && ell)) ? rowptr[row[u]] : 0;
        e1[u] = (ok && !(VEC == 4 && ell)) ? rowptr[row[u] + 1] : 0;
    }
    float acc[RPT][VEC];
#pragma unroll
    for (int u = 0; u < RPT; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[u][k] = 0.0f;
    // the addends are requested now, together with the index loads, not after the gather loop: one dependent memory phase
    // less at the end of every launch (each thread reads its own p / q element before it writes out: aliasing is fine)
    float pv[RPT][VEC], qv[RPT][VEC];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        if constexpr (VEC == 4) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (p) a = *reinterpret_cast<const float4*>(p + row[u] * P.ldp + ch);
            if (q) b = *reinterpret_cast<const float4*>(q + row[u] * P.ldq + ch);
            pv[u][0] = a.x; pv[u][1] = a.y; pv[u][2] = a.z; pv[u][3] = a.w;
            qv[u][0] = b.x; qv[u][1] = b.y; qv[u][2] = b.z; qv[u][3] = b.w;
        } else {
            pv[u][0] = p ? p[row[u] * P.ldp + ch] : 0.0f;
            qv[u][0] = q ? q[row[u] * P.ldq + ch] : 0.0f;
        }
    }
    if constexpr (VEC == 4) {
        if (ell) {
            // the first four edges of a row come as two 16-byte vectors (qt_edges_norm): no row pointer on the way to the
            // gathers, fully regular index loads; only rows flagged with more than four edges go on to the CSR loop below
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                int4 c4 = ell[2 * row[u]];
                const int4 wb = ell[2 * row[u] + 1];
                const bool more4 = c4.w < 0;
                if (more4) c4.w = ~c4.w;
                const float4 f0 = *reinterpret_cast<const float4*>(x + (int64_t)c4.x * ldx + ch);
                const float4 f1 = *reinterpret_cast<const float4*>(x + (int64_t)c4.y * ldx + ch);
                const float4 f2 = *reinterpret_cast<const float4*>(x + (int64_t)c4.z * ldx + ch);
                const float4 f3 = *reinterpret_cast<const float4*>(x + (int64_t)c4.w * ldx + ch);
                const float w0 = __int_as_float(wb.x), w1 = __int_as_float(wb.y), w2 = __int_as_float(wb.z), w3 = __int_as_float(wb.w);
                acc[u][0] += w0 * f0.x; acc[u][1] += w0 * f0.y; acc[u][2] += w0 * f0.z; acc[u][3] += w0 * f0.w;
                acc[u][0] += w1 * f1.x; acc[u][1] += w1 * f1.y; acc[u][2] += w1 * f1.z; acc[u][3] += w1 * f1.w;
                acc[u][0] += w2 * f2.x; acc[u][1] += w2 * f2.y; acc[u][2] += w2 * f2.z; acc[u][3] += w2 * f2.w;
                acc[u][0] += w3 * f3.x; acc[u][1] += w3 * f3.y; acc[u][2] += w3 * f3.z; acc[u][3] += w3 * f3.w;
                if (more4 && rp * RPT + u < rows) {
                    e0[u] = rowptr[row[u]] + 4;
                    e1[u] = rowptr[row[u] + 1];
                }
            }
        }
    }
    // 4 edges per trip: the index/weight loads, then the neighbour gathers, are independent and stay in flight together
    // (quadtree rows have ~4 neighbours, so most rows finish in the first trip, which is issued for all RPT rows at once).
    // Tried and rejected (round 1): staging each 64-row run of the reversed-Morton node order in LDS so that the ~87 %
    // internal neighbours are LDS reads -- 12.2 us vs 11.2 us per launch at N = 1.2e5, C = 20.
    auto trip = [&](const int (&eb)[RPT], bool (&more)[RPT]) {
        int cj[RPT][EPT];
        float w[RPT][EPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u)
#pragma unroll
            for (int v = 0; v < EPT; ++v) {
                const bool ok = eb[u] + v < e1[u];
                cj[u][v] = ok ? col[eb[u] + v] : (int)row[u];
                w[u][v] = ok ? nrm[eb[u] + v] : 0.0f;
            }
        if constexpr (VEC == 4) {
            // gathers in groups of 4 edges: the first group always (slots beyond the row's edges re-read the row itself with
            // weight 0), a further group only for the lanes whose row reaches it -- one branch per group, so a row with 4
            // neighbours costs the texture addresser 4 gathers, not EPT, while the index loads of all EPT slots (issued
            // above) stay off the critical path of the rows with many neighbours
#pragma unroll
            for (int v0 = 0; v0 < EPT; v0 += 4) {
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    if (v0 > 0 && eb[u] + v0 >= e1[u]) continue;
                    float4 f[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) f[v] = *reinterpret_cast<const float4*>(x + (int64_t)cj[u][v0 + v] * ldx + ch);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        acc[u][0] += w[u][v0 + v] * f[v].x; acc[u][1] += w[u][v0 + v] * f[v].y;
                        acc[u][2] += w[u][v0 + v] * f[v].z; acc[u][3] += w[u][v0 + v] * f[v].w;
                    }
                }
            }
        } else {
            float f[RPT][EPT];
#pragma unroll
            for (int u = 0; u < RPT; ++u)
#pragma unroll
                for (int v = 0; v < EPT; ++v) f[u][v] = x[(int64_t)cj[u][v] * ldx + ch];
#pragma unroll
            for (int u = 0; u < RPT; ++u)
#pragma unroll
                for (int v = 0; v < EPT; ++v) acc[u][0] += w[u][v] * f[u][v];
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) more[u] = eb[u] + EPT < e1[u];
    };
    int eb[RPT];
    bool more[RPT];
    bool any = false;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        eb[u] = e0[u];
        any |= e0[u] < e1[u];
    }
    while (any) {
        trip(eb, more);
        any = false;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            eb[u] = more[u] ? eb[u] + EPT : e1[u];       // a finished row keeps an empty edge range
            any |= more[u];
        }
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        if (rp * RPT + u >= rows) continue;
        const int64_t o = row[u] * C + ch;
        float r[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            r[k] = alpha * acc[u][k];
            if (p) r[k] += beta * pv[u][k];
            if (q) r[k] += gamma * qv[u][k];
        }
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(out + o) = make_float4(r[0], r[1], r[2], r[3]);
        } else {
            out[o] = r[0];
        }
    }
}

// One-column message aggregate with strided operands and an optional output epilogue: the Clenshaw recurrence of a
// ChebConv with ONE output channel (the decoder's fc_out2, model/seq2seq.py:121) after its coefficient columns have been
// applied -- u = z [w_0 w_1 w_2] first, then y = u_0 + L^ (u_1 + 2 L^ u_2) - u_2 on single columns of the (N, 4) matrix u:
// the two propagations move 4 bytes per row and neighbour instead of the 64-byte rows of z.  Thread = row.
struct Spmm1Args {
    const int32_t* rowptr;
    const int32_t* col;
    const float* nrm;
    const int4* ell;
    int Ncap;
    const int32_t* n_dev;
    const float *x, *p, *q;
    int ldx, ldp, ldq;
    float alpha, beta, gamma;
    float* out;
    int ldo, pad4;          // pad4: the row is written as (v, 0, 0, 0)
    int act;                // QT_ACT_NONE or QT_ACT_TANH_RES: v = tanh(drop * v) + res
    const float* res;
    int ldr;
    const float* drop;
};
__global__ __launch_bounds__(64) void k_spmm1(Spmm1Args a) {
    const int rows = qt_rows(a.n_dev, a.Ncap);
    const int nblk = (rows + 63) >> 6, chunk = (nblk + 7) >> 3;           // XCD-wise row ranges, as k_spmm
    const int bid = blockIdx.x;
    if ((bid >> 3) >= chunk) return;
    const int64_t row = (int64_t)((bid & 7) * chunk + (bid >> 3)) * 64 + threadIdx.x;
    if (row >= rows) return;
    const float pv = a.p ? a.p[row * a.ldp] : 0.0f, qv = a.q ? a.q[row * a.ldq] : 0.0f;
    const float* __restrict__ x = a.x;
    float acc = 0.0f;
    int e0, e1;
    if (a.ell) {
        int4 c4 = a.ell[2 * row];
        const int4 wb = a.ell[2 * row + 1];
        const bool more4 = c4.w < 0;
        if (more4) c4.w = ~c4.w;
        const float f0 = x[(int64_t)c4.x * a.ldx], f1 = x[(int64_t)c4.y * a.ldx], f2 = x[(int64_t)c4.z * a.ldx],
                    f3 = x[(int64_t)c4.w * a.ldx];
        acc += __int_as_float(wb.x) * f0;
        acc += __int_as_float(wb.y) * f1;
        acc += __int_as_float(wb.z) * f2;
        acc += __int_as_float(wb.w) * f3;
        e0 = e1 = 0;
        if (more4) {
            e0 = a.rowptr[row] + 4;
            e1 = a.rowptr[row + 1];
        }
    } else {
        e0 = a.rowptr[row];
        e1 = a.rowptr[row + 1];
    }
    for (int eb = e0; eb < e1; eb += 4) {
        int cj[4];
        float w[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const bool ok = eb + v < e1;
            cj[v] = ok ? a.col[eb + v] : (int)row;
            w[v] = ok ? a.nrm[eb + v] : 0.0f;
        }
        float f[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) f[v] = x[(int64_t)cj[v] * a.ldx];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc += w[v] * f[v];
    }
    float v = a.alpha * acc;
    if (a.p) v += a.beta * pv;
    if (a.q) v += a.gamma * qv;
    if (a.act == QT_ACT_TANH_RES) v = tanhf((a.drop ? a.drop[row] : 1.0f) * v) + a.res[row * a.ldr];
    if (a.pad4)
        *reinterpret_cast<float4*>(a.out + row * a.ldo) = make_float4(v, 0.0f, 0.0f, 0.0f);
    else
        a.out[row * a.ldo] = v;
}

// ------------------------------------------------------------------ tiled GEMM
// Node-feature operand made of Ka planes plus an optional (N, Ks) block.  A plane is one (N, Ca) matrix or two matrices side
// by side, (N, Ca) | (N, Cab): the recurrent cells feed Z = [X | H] without ever concatenating it -- rows of 64 bytes (H)
// and 16 bytes (X) also keep every 4-lane group of a gather inside one row, which rows of 80 bytes do not.
struct PlaneSrc {
    const float* a0;        // plane 0, part a (N, Ca)
    const float* a_rest;    // planes 1 .. Ka-1, part a (Ka-1, N, Ca)
    const float* a0b;       // part b of the same planes: (N, Cab) and (Ka-1, N, Cab); Cab == 0: none
    const float* a_restb;
    const float* S;
    int Ka, Ca, Cab, Ks, N;
    int lda0, lda0b;        // row strides of plane 0 (column views of wider matrices are passed as they are)
    int sm;                 // planes 1 .. Ka-1 are stored SLICE-major: (plane, 4-channel slice, N, 4) -- the layout the clip-resident
                            // recurrence writes (consecutive rows of a slice are contiguous: coalesced stores there, and a quad of a
                            // row is reached at slice base + row * 4 here)
};

struct GemmArgs {
    PlaneSrc A;        // forward: left operand rows = nodes; wgrad: transposed use
    const float* B;    // forward: W (K, NB); wgrad: G (N, NB)
    const float* BT;   // forward, optional: W^T (NB, K) -- staged with straight float4 copies instead of a transposing scatter
    int M, K, NB;      // output M x NB, reduction K
    // forward epilogue
    int Kb, Cb, act;
    const float* res;
    int res_stride;
    const float* drop;
    float* out;
    float* outb;        // forward: second column part of every output plane, (Kb, M, Cbb); Cbb == 0: none
    int Cbb;
    int out_sm;         // output planes 1 .. Kb-1 slice-major (plane_piece)
    // k_gemm_skinny<64> with NB = 16: a second product in the epilogue, post_out (M, 4) = [act(out) | 1 0 0 0] @ post_W (NB + 4, 4)
    // (the decoder head: fc_out1's 16 channels -> the three coefficient columns of fc_out2, seq2seq.py:115-121)
    const float* post_W;
    float* post_out;
    int64_t row0_step;  // wgrad: rows per block
    const int32_t* n_dev;  // valid node rows on the device (NULL: A.N)
    int accumulate;        // wgrad: add into part instead of overwriting (sums several uses of one weight)
    // gate GEMM with the LSTM cell as its epilogue (k_gemm_fwd<2, 128, true>): the (N, 4h) pre-activations never leave LDS
    const float* Cprev;
    const float* wc;
    const float* bias;
    const float* ln;
    int ld_c, h;
    float *O, *Hn, *Cn, *gates;
    // grouped use (qt_proj_group): blockIdx.z = group; plane 0, the weight and the output of group z start gsA / gsB / gsO
    // floats after those of group z - 1.  ldo: row stride of the output plane (0 = Cb; a column block of a wider matrix)
    int ldo, zrev;
    int64_t gsA, gsB, gsO;
#ifdef QT_GEMM_TIMING
    long long* dbg;
#endif
};
#ifdef QT_GEMM_TIMING
#define QT_STAMP(i) do { if (g.dbg && threadIdx.x == 0) g.dbg[(int64_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define QT_STAMP(i) do {} while (0)
#endif

// ---- fp32 MFMA tiles (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD).
// Operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; accumulator register r of lane l is C[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
// The two k slots of one MFMA may be ANY two reduction indices as long as A and B agree, which is what lets a
// lane fetch its A operand as one float4 (4 consecutive k of its own row): in step (j, i) lane half h feeds
// k = 8 j + 4 h + i.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BM = 128;       // block rows (4 waves x 32)
constexpr int BN = 64;        // block columns (2 MFMA tiles per wave)
constexpr int MAXQ = 128;     // quads (4 consecutive k) in the reduction dimension

// Pointers that went through the LDS quad table lose their address space: hipcc then emits flat_load, and flat loads
// force `s_waitcnt vmcnt(0) lgkmcnt(0)` at every use (they may return out of order), which serialised the whole
// operand stream.  Loading through an explicit global (address space 1) pointer restores counted vmcnt waits.
typedef float qt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(const float* p) {
    const qt_v4f v = *(const __attribute__((address_space(1))) qt_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// float4 piece (plane pl, row i, channel ch) of an OUTPUT plane set of width C: row-major (planes, M, C) -- or, with sm, planes
// 1.. slice-major (C / 4, M, 4), the layout the clip-resident Clenshaw launch reads with coalesced loads (plane 0 stays row-major:
// it becomes the gradient matrix of the layer's input)
__device__ __forceinline__ float* plane_piece(float* base, int pl, int64_t i, int ch, int C, int64_t M, int sm, int ld) {
    if (sm && pl > 0) return base + (((int64_t)pl * (C >> 2) + (ch >> 2)) * M + i) * 4;
    return base + (int64_t)pl * M * C + i * ld + ch;
}

// Quad table: quad Q of a node row lives at qptr[Q] + row * qstr[Q] (plane Q*4/Ca of the operand, or S).
__device__ __forceinline__ void build_quad_table(const PlaneSrc& A, const float** qptr, int* qstr, int nquad) {
    for (int Q = threadIdx.x; Q < nquad; Q += 256) {
        const int ct = A.Ca + A.Cab;
        const int k = 4 * Q, kc = A.Ka * ct;
        if (k < kc) {
            const int pl = k / ct, c = k - pl * ct;
            if (c < A.Ca) {
                if (pl > 0 && A.sm) {
                    qptr[Q] = A.a_rest + ((int64_t)(pl - 1) * (A.Ca / 4) + c / 4) * A.N * 4;
                    qstr[Q] = 4;
                } else {
                    qptr[Q] = (pl == 0 ? A.a0 : A.a_rest + (int64_t)(pl - 1) * A.N * A.Ca) + c;
                    qstr[Q] = pl == 0 ? A.lda0 : A.Ca;
                }
            } else {
                if (pl > 0 && A.sm) {
                    qptr[Q] = A.a_restb + ((int64_t)(pl - 1) * (A.Cab / 4) + (c - A.Ca) / 4) * A.N * 4;
                    qstr[Q] = 4;
                } else {
                    qptr[Q] = (pl == 0 ? A.a0b : A.a_restb + (int64_t)(pl - 1) * A.N * A.Cab) + (c - A.Ca);
                    qstr[Q] = pl == 0 ? A.lda0b : A.Cab;
                }
            }
        } else {
            qptr[Q] = A.S + (k - kc);
            qstr[Q] = A.Ks;
        }
    }
}

// MODE 0: out planes = act(A @ W).  Block = 128 node rows x (32 NT) output columns, wave w owns rows [32w, 32w+32).
// A fragments go global -> VGPR directly (float4 per lane and k-quad); only W is staged in LDS (KWT x 32 NT floats).
// NT = 2 for NB <= 64 (gate GEMM), NT = 4 for wide outputs (the data gradient, NB = K*C) so A is read only once.
template <int NT, int KWT, int CELL = 0>     // CELL: 0 = plain epilogue, else the lanes per node (h / 4) of the fused LSTM cell
#ifndef QT_GEMM_OCC
#define QT_GEMM_OCC 4
#endif
#ifndef QT_GEMM_OCC3
#define QT_GEMM_OCC3 3
#endif
#ifndef QT_GEMM_OCC4C
#define QT_GEMM_OCC4C 2
#endif
__global__ __launch_bounds__(256, (NT == 4 && CELL != 0) ? QT_GEMM_OCC4C : (NT == 3 ? QT_GEMM_OCC3 : QT_GEMM_OCC)) void k_gemm_fwd(GemmArgs g) {   // 4 workgroups per CU: all N/128 blocks of the
                                                                      // bench shape are resident at once (<= 128 registers)
    constexpr int BNT = 32 * NT;
    constexpr int PITCH = KWT + 4;      // == 4 (mod 64) floats: the 16 lanes of a ds_read_b128 group hit distinct banks
    // W chunk TRANSPOSED in LDS, Bt[column][k]: a lane's four B operands of one k-quad are one ds_read_b128
    // (measured: with one ds_read_b32 per MFMA the kernel ran at half the MFMA rate)
    __shared__ __attribute__((aligned(16))) float Bt[BNT * PITCH];
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * BM;
    const int j0 = blockIdx.y * BNT;
    const int64_t rows = qt_rows(g.n_dev, g.M);      // g.M stays the plane stride (capacity)
    if (i0 >= rows) return;
    QT_STAMP(0);
    if (gridDim.z > 1) {
        const int z = g.zrev ? (int)gridDim.z - 1 - (int)blockIdx.z : (int)blockIdx.z;     // (zrev: groups from the last to the first)
        g.A.a0 += z * g.gsA;
        if (g.A.a_rest) g.A.a_rest += z * g.gsA;
        if (g.B) g.B += z * g.gsB;
        if (g.BT) g.BT += z * g.gsB;
        g.out += z * g.gsO;
    }
    const int nquad = g.K >> 2;
    build_quad_table(g.A, qptr, qstr, nquad);
    const int64_t my_row = i0 + wave * 32 + l32;
    const bool row_ok = my_row < rows;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    // fused cell: this thread's previous cell states (one node per epilogue pass) are requested now, before the MFMA loop,
    // so the epilogue does not start with a dependent memory round trip
    constexpr int NPASS = CELL != 0 ? BM / (256 / (CELL != 0 ? CELL : 1)) : 1;
    float4 cpre[NPASS];
    if constexpr (CELL != 0) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int64_t node = i0 + ps * (256 / CELL) + t / CELL;
            cpre[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (node < rows && g.Cprev) cpre[ps] = *reinterpret_cast<const float4*>(g.Cprev + node * g.ld_c + (t % CELL) * 4);
        }
    }
    for (int k0 = 0; k0 < g.K; k0 += KWT) {
        const int kn = min(KWT, g.K - k0);           // multiple of 4
        __syncthreads();                              // table ready / previous pass done with Bs
        QT_STAMP(1);
        // W chunk -> LDS first (small, L2 resident) ...
        if (g.BT) {
            // ... from W^T: a column's k run is contiguous in memory and in LDS (conflict-free 16-byte stores)
            const int kqn = kn >> 2;
            for (int e = t; e < BNT * kqn; e += 256) {
                const int c = e / kqn, kq = e - c * kqn;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j0 + c < g.NB) w = *reinterpret_cast<const float4*>(g.BT + (int64_t)(j0 + c) * g.K + k0 + 4 * kq);
                *reinterpret_cast<float4*>(&Bt[c * PITCH + 4 * kq]) = w;
            }
        } else {
            for (int e = t; e < kn * (BNT / 4); e += 256) {
                const int kb = e / (BNT / 4), jq = (e % (BNT / 4)) * 4;
                float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j0 + jq < g.NB) w = *reinterpret_cast<const float4*>(g.B + (int64_t)(k0 + kb) * g.NB + j0 + jq);
                Bt[(jq + 0) * PITCH + kb] = w.x;
                Bt[(jq + 1) * PITCH + kb] = w.y;
                Bt[(jq + 2) * PITCH + kb] = w.z;
                Bt[(jq + 3) * PITCH + kb] = w.w;
            }
        }
        for (int e = t; e < BNT * ((KWT - kn) / 4); e += 256) {          // zero the k tail of a short last pass
            const int c = e / ((KWT - kn) / 4), kq = kn + (e % ((KWT - kn) / 4)) * 4;
            *reinterpret_cast<float4*>(&Bt[c * PITCH + kq]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        QT_STAMP(2);
        // ... then the MFMA stream.  The A quads (quad 2 j + half of this lane's row) come straight from global memory
        // through a 4-deep register ring loaded four k-groups ahead; the loop is a plain runtime loop with NO branch
        // around the MFMAs (conditionals there made hipcc shuttle the accumulators between VGPRs and AGPRs: 1088
        // v_accvgpr moves and a vmcnt(0) per group, 2.75x slower than the MFMA rate).
        const int q0 = k0 >> 2, qn = kn >> 2, nj = (kn + 7) >> 3;
        auto ldq = [&](int j) {
            const int q = 2 * j + half;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && q < qn) r = gload4(qptr[q0 + q] + my_row * qstr[q0 + q]);
            return r;
        };
        float4 a0 = ldq(0), a1 = ldq(1), a2 = ldq(2), a3 = ldq(3);
        for (int j = 0; j < nj; ++j) {
            const float4 a = a0;
            a0 = a1; a1 = a2; a2 = a3;
            a3 = ldq(j + 4);
            float4 bq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bq[nt] = *reinterpret_cast<const float4*>(&Bt[(nt * 32 + l32) * PITCH + 8 * j + 4 * half]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[nt].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[nt].y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[nt].z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[nt].w, acc[nt], 0, 0, 0);
            }
        }
    }
    QT_STAMP(3);
    float* Cs = Bt;                              // 128 rows x 64 columns per pass
    if constexpr (CELL != 0) {
        // LSTM epilogue: h / 4 adjacent lanes own a node, as in k_lstm_fwd (h = 8, 16 with NT = 2; h = 32 with NT = 4: all
        // four gates of a node sit in this block's 32 NT columns).  The gate tile goes through LDS 256 / (h/4) rows at a
        // time with a row pitch of 5 h floats: the rows a 16-lane ds_read_b128 phase touches then start h banks apart.
        // Same arithmetic, in the same order, as qt_dense followed by qt_lstm_fwd.
        using namespace qtcell;
        constexpr int lpn = CELL;                // 2, 4 or 8 lanes per node
        constexpr int h = 4 * lpn;
        constexpr int CP = 5 * h;
        constexpr int RP = 256 / lpn;            // rows per pass: 128, 64 or 32
        static_assert(4 * h <= 32 * NT && BNT * PITCH >= RP * CP, "gate tile does not fit");
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += RP) {
            __syncthreads();
            if (wave * 32 >= r0 && wave * 32 < r0 + RP) {
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (u * 32 < 4 * h)      // (h = 8: the second 32-column tile is padding)
                            Cs[(wave * 32 - r0 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + u * 32 + l32] = acc[u][r];
            }
            __syncthreads();
            const int row = t / lpn, j0 = (t - row * lpn) * 4;
            const int64_t node = i0 + r0 + row;
            const bool ok = node < rows;
            const float* cs = Cs + row * CP + j0;
            const F4 gi = ld4(cs), gf = ld4(cs + h), gc = ld4(cs + 2 * h), go = ld4(cs + 3 * h);
            const float4 c4 = cpre[r0 / RP];
            const F4 cp = {{c4.x, c4.y, c4.z, c4.w}};
            const CellOut r = cell_forward<lpn>(gi, gf, gc, go, cp, g.wc, g.bias, g.ln, h, j0);
            if (ok) {
                if (g.O) st4(g.O + node * h + j0, r.Og);
                st4(g.Hn + node * h + j0, r.hn);
                st4(g.Cn + node * h + j0, r.cn);
                float* gs = g.gates + node * 4 * h + j0;
                st4(gs, r.I);
                st4(gs + h, r.F);
                st4(gs + 2 * h, r.T);
                st4(gs + 3 * h, r.Og);
            }
        }
        return;
    }
    // Epilogue: an MFMA accumulator holds one COLUMN per lane; staging the tile in LDS (the W buffer is free now) lets
    // every thread write float4 pieces of output ROWS instead (4x fewer, 16-byte wide, row-contiguous stores).
    static_assert(BNT * PITCH >= BM * 64, "LDS staging tile does not fit in the W buffer");
#pragma unroll
    for (int h2 = 0; h2 < (NT + 1) / 2; ++h2) {        // two 32-column MFMA tiles per pass (the last pass of an odd NT: one)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nt = 2 * h2 + u;
            if (nt < NT)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + u * 32 + l32] = acc[nt][r];
        }
        __syncthreads();
        QT_STAMP(4);
#pragma unroll
        for (int u = 0; u < BM * 16 / 256; ++u) {
            const int e = t + 256 * u;
            const int row = e >> 4, c4 = (e & 15) * 4;
            const int64_t i = i0 + row;
            const int j = j0 + h2 * 64 + c4;
            if (i >= rows || j >= g.NB || h2 * 64 + c4 >= BNT) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * 64 + c4]);
            if (g.act == QT_ACT_RELU) {
                v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            }
            if (g.act == QT_ACT_TANH_RES) {
                const float d = g.drop ? g.drop[i] : 1.0f, rs = g.res[i * g.res_stride];
                v.x = tanhf(d * v.x) + rs; v.y = tanhf(d * v.y) + rs; v.z = tanhf(d * v.z) + rs; v.w = tanhf(d * v.w) + rs;
            }
            const int ct = g.Cb + g.Cbb;
            const int pl = j / ct, ch = j - pl * ct;              // Cb, Cbb % 4 == 0: a float4 never straddles two parts
            if (ch < g.Cb)
                *reinterpret_cast<float4*>(plane_piece(g.out, pl, i, ch, g.Cb, g.M, g.out_sm, g.ldo ? g.ldo : g.Cb)) = v;
            else
                *reinterpret_cast<float4*>(plane_piece(g.outb, pl, i, ch - g.Cb, g.Cbb, g.M, g.out_sm, g.Cbb)) = v;
        }
    }
    QT_STAMP(5);
}

// ---- persistent gate GEMM + LSTM cell (hidden 8 / 16): one 512-thread workgroup per CU, W staged ONCE, no workgroup barrier
// after that.  The unit of work is a WAVE's 32 node rows x all 4h gate columns: the wave streams its A quads global -> VGPR
// through a 4-deep ring that already holds the next unit's first quads when the current unit's epilogue starts, runs the
// MFMA chain, parks the accumulators in its OWN staging rows in LDS and computes the cell for those 32 nodes itself (h / 4
// lanes per node, as k_gemm_fwd's fused epilogue: same arithmetic in the same order, bit-identical results).  Two waves
// share a SIMD, so one wave's epilogue (VALU, LDS, stores) runs beside the other's MFMA chain.  Against the one-tile
// workgroups of k_gemm_fwd<2, 128, 4> this removes the per-tile W staging (4 us of 21 at the bench shape), the four
// workgroup barriers per tile, and the serial memory -> MFMA -> store phases of a tile.
// Work split: workgroup b owns the contiguous units [U b / G, U (b + 1) / G) of the U = ceil(rows / 32) units (valid rows
// read on the device), its wave w takes every 8th of them.
constexpr int GATE_P_MAXK = 256, GATE_P_MAXPITCH = GATE_P_MAXK + 8;
#ifndef QT_GATE_STAGGER
#define QT_GATE_STAGGER 0
#endif
template <int NT, int LPN, int R>
__global__ __launch_bounds__(512, 2) void k_gate_cell_p(GemmArgs g, int pitch) {
    using namespace qtcell;
    constexpr int BNT = 32 * NT, h = 4 * LPN, CP = 5 * h, NPW = 64 / LPN, NPASS = 32 / NPW;
    static_assert(4 * h == BNT, "the gate columns fill the MFMA tiles exactly");
    // static LDS (a single workgroup may declare up to 160 KiB on gfx950; dynamic LDS beyond 64 KiB was refused at launch)
    __shared__ __attribute__((aligned(16))) float Bt[BNT * GATE_P_MAXPITCH];   // W^T: [BNT][pitch], pitch / 4 odd -> conflict-free ds_read_b128
    __shared__ __attribute__((aligned(16))) float Cst[8 * 32 * CP];
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    float* Cs = Cst + wave * (32 * CP);                 // this wave's staging rows
    const int rows = qt_rows(g.n_dev, g.M);
    const int nunits = (rows + 31) >> 5;
    const int u0 = (int)((int64_t)nunits * blockIdx.x / gridDim.x), u1 = (int)((int64_t)nunits * (blockIdx.x + 1) / gridDim.x);
    if (u0 >= u1) return;
    QT_STAMP(0);
    const int nquad = g.K >> 2;
    build_quad_table(g.A, qptr, qstr, nquad);
    if (g.BT) {
        for (int e = t; e < BNT * nquad; e += 512) {
            const int c = e / nquad, kq = e - c * nquad;
            *reinterpret_cast<float4*>(&Bt[c * pitch + 4 * kq]) = *reinterpret_cast<const float4*>(g.BT + (int64_t)c * g.K + 4 * kq);
        }
    } else {
        for (int e = t; e < g.K * (BNT / 4); e += 512) {
            const int kb = e / (BNT / 4), jq = (e % (BNT / 4)) * 4;
            const float4 w = *reinterpret_cast<const float4*>(g.B + (int64_t)kb * BNT + jq);
            Bt[(jq + 0) * pitch + kb] = w.x;
            Bt[(jq + 1) * pitch + kb] = w.y;
            Bt[(jq + 2) * pitch + kb] = w.z;
            Bt[(jq + 3) * pitch + kb] = w.w;
        }
    }
    if (g.K & 4)                                          // an odd quad count: the last k-group's upper half reads zeros
        for (int c = t; c < BNT; c += 512) *reinterpret_cast<float4*>(&Bt[c * pitch + g.K]) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();                                      // the only workgroup barrier
    QT_STAMP(1);
    int unit = u0 + wave;
    if (unit >= u1) return;
    const int nj = (g.K + 7) >> 3;
    // Every A load is UNCONDITIONAL (row and quad clamped to valid ones, the value zeroed by a select): a load inside a
    // branch makes hipcc's s_waitcnt bookkeeping fall back to draining the whole queue at the next use, which serialised
    // the stream (41 us per launch at the bench shape, whatever the ring depth).
    const int64_t last_row = rows - 1;
    auto ldq = [&](int64_t row, bool ok, int j) {
        const int q = 2 * j + half;
        const bool use = ok && q < nquad;
        const int qc = q < nquad ? q : 0;
        const float4 r = gload4(qptr[qc] + (row <= last_row ? row : last_row) * qstr[qc]);
        return make_float4(use ? r.x : 0.f, use ? r.y : 0.f, use ? r.z : 0.f, use ? r.w : 0.f);
    };
    int64_t my_row = (int64_t)unit * 32 + l32;
    bool row_ok = my_row < rows;
    // The A operand is streamed once and shared with no other wave: it goes global -> VGPR, and what bounds the stream is the
    // bytes a CU keeps in flight (8 waves x 4 quads of 1 KiB = 32 KiB ran at 2.5 TB/s).  R > 0: the ring holds a WHOLE unit
    // (nj <= R steps, the j loop fully unrolled so that ring[j] is a fixed register): step j consumes ring[j] and at once
    // requests the next unit's quad j into it, so a wave always has ~nj KiB in flight, across the epilogue too.
    // R == 0 (any nj): the 4-deep rotating ring of k_gemm_fwd.
    float4 ring[R > 0 ? R : 4];
#pragma unroll
    for (int j = 0; j < (R > 0 ? R : 4); ++j) ring[j] = ldq(my_row, row_ok, j);
    const int nl = lane / LPN, j0 = (lane - nl * LPN) * 4;
    const CellParams cpar = cell_params(g.wc, g.bias, g.ln, h, j0);      // in registers for the whole launch
    // Stagger: the two waves of a SIMD (w and w + 4) run the same program; started together they sit in their MFMA chains
    // together (matrix pipe shared: 2 x 6.7k cycles) and then in their epilogues together (VALU issue shared: the cell's
    // sigmoids / tanhs are ~6k cycles per unit) -- in-kernel stamps showed 7.7 + 7.5 us per unit that way.  Waves 4..7 start
    // half a unit late, so one wave's epilogue runs beside the other's MFMA chain.
    if (wave >= 4) {
#pragma unroll 1
        for (int i = 0; i < QT_GATE_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);       // 64 x 64 clocks per trip
    }
    while (true) {
        // this unit's previous cell states (one node per epilogue pass and lane group): requested before the MFMA chain
        float4 cpre[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int64_t node = (int64_t)unit * 32 + ps * NPW + nl;
            cpre[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (node < rows && g.Cprev) cpre[ps] = *reinterpret_cast<const float4*>(g.Cprev + node * g.ld_c + j0);
        }
        const int nxt = unit + 8;
        const bool has_next = nxt < u1;
        const int64_t nrow = (int64_t)nxt * 32 + l32;
        const bool nok = has_next && nrow < rows;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
        auto step = [&](const float4& a, int j) {
            float4 bq[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bq[nt] = *reinterpret_cast<const float4*>(&Bt[(nt * 32 + l32) * pitch + 8 * j + 4 * half]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#ifdef QT_EXP_NOMFMA
                acc[nt][0] += a.x * bq[nt].x + a.y * bq[nt].y + a.z * bq[nt].z + a.w * bq[nt].w;
#else
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[nt].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[nt].y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[nt].z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[nt].w, acc[nt], 0, 0, 0);
#endif
            }
        };
        if constexpr (R > 0) {
#pragma unroll
            for (int j = 0; j < R; ++j) {     // nj == R (the host picks the instance): straight-line code, no branch
                const float4 a = ring[j];
                ring[j] = ldq(nrow, nok, j);
                step(a, j);
            }
        } else {
            for (int j = 0; j < nj; ++j) {
                const float4 a = ring[0];
                ring[0] = ring[1]; ring[1] = ring[2]; ring[2] = ring[3];
                ring[3] = ldq(my_row, row_ok, j + 4);
                step(a, j);
            }
            // the next unit's first quads fly during this unit's epilogue
#pragma unroll
            for (int j = 0; j < 4; ++j) ring[j] = ldq(nrow, nok, j);
        }
        // epilogue: accumulator columns -> this wave's staging rows -> h / 4 lanes per node
        if (unit == u0) QT_STAMP(2);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) Cs[((r & 3) + 8 * (r >> 2) + 4 * half) * CP + u * 32 + l32] = acc[u][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int row = ps * NPW + nl;
            const int64_t node = (int64_t)unit * 32 + row;
            const float* cs = Cs + row * CP + j0;
            const F4 gi = ld4(cs), gf = ld4(cs + h), gc = ld4(cs + 2 * h), go = ld4(cs + 3 * h);
            const F4 cp = {{cpre[ps].x, cpre[ps].y, cpre[ps].z, cpre[ps].w}};
#ifdef QT_EXP_NOCELL
            CellOut r;
            r.I = gi; r.F = gf; r.T = gc; r.Og = go; r.hn = cp; r.cn = cp;
#else
            const CellOut r = cell_forward<LPN>(gi, gf, gc, go, cp, cpar, h);
#endif
            if (node < rows) {
                if (g.O) st4(g.O + node * h + j0, r.Og);
                st4(g.Hn + node * h + j0, r.hn);
                st4(g.Cn + node * h + j0, r.cn);
                float* gs = g.gates + node * 4 * h + j0;
                st4(gs, r.I);
                st4(gs + h, r.F);
                st4(gs + 2 * h, r.T);
                st4(gs + 3 * h, r.Og);
            }
        }
        if (unit == u0) QT_STAMP(3);
        if (!has_next) break;
        unit = nxt; my_row = nrow; row_ok = nok;
    }
    QT_STAMP(4);
}

// ---- skinny shapes: few output columns (the decoder head: 16 or 4) or a short reduction (its data gradients: K = 16 or 4).
// The MFMA kernel's fixed costs (W staging, barriers, the LDS round trip of the epilogue: ~15 us) dwarf such a product; here a
// wave owns 64 rows x 4 output columns, reads its A quads straight from global memory and the matching 4 x 4 block of W
// through scalar loads (wave uniform), and accumulates with plain fp32 FMAs -- the VALU has the fp32 MFMA's FLOP rate.
// RPB = 64: the 4 waves of a workgroup take 4 column quads of the same 64 rows (their A loads meet in L1); RPB = 256: one
// column quad in all (NB = 4).
template <int RPB>
__global__ __launch_bounds__(256) void k_gemm_skinny(GemmArgs g) {
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    __shared__ float hs[RPB == 64 ? 64 * 17 : 1];       // (post product: the 64 x 16 output tile, pitch 17)
    const int t = threadIdx.x;
    const int64_t rows = qt_rows(g.n_dev, g.M);
    const int64_t row0 = (int64_t)blockIdx.x * RPB;
    if (row0 >= rows) return;
    const int nquad = g.K >> 2;
    build_quad_table(g.A, qptr, qstr, nquad);
    __syncthreads();
    const int cq = __builtin_amdgcn_readfirstlane(RPB == 64 ? (int)blockIdx.y * 4 + (t >> 6) : (int)blockIdx.y);
    const int j = cq * 4;
    if (j >= g.NB) return;
    const int64_t row = row0 + (RPB == 64 ? (t & 63) : t);
    const bool ok = row < rows;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* __restrict__ Wc = g.B + j;
    auto step = [&](const float4& a, int Q) {
        const float4 w0 = *reinterpret_cast<const float4*>(Wc + (int64_t)(4 * Q + 0) * g.NB);
        const float4 w1 = *reinterpret_cast<const float4*>(Wc + (int64_t)(4 * Q + 1) * g.NB);
        const float4 w2 = *reinterpret_cast<const float4*>(Wc + (int64_t)(4 * Q + 2) * g.NB);
        const float4 w3 = *reinterpret_cast<const float4*>(Wc + (int64_t)(4 * Q + 3) * g.NB);
        acc.x = fmaf(a.x, w0.x, acc.x); acc.y = fmaf(a.x, w0.y, acc.y); acc.z = fmaf(a.x, w0.z, acc.z); acc.w = fmaf(a.x, w0.w, acc.w);
        acc.x = fmaf(a.y, w1.x, acc.x); acc.y = fmaf(a.y, w1.y, acc.y); acc.z = fmaf(a.y, w1.z, acc.z); acc.w = fmaf(a.y, w1.w, acc.w);
        acc.x = fmaf(a.z, w2.x, acc.x); acc.y = fmaf(a.z, w2.y, acc.y); acc.z = fmaf(a.z, w2.z, acc.z); acc.w = fmaf(a.z, w2.w, acc.w);
        acc.x = fmaf(a.w, w3.x, acc.x); acc.y = fmaf(a.w, w3.y, acc.y); acc.z = fmaf(a.w, w3.z, acc.z); acc.w = fmaf(a.w, w3.w, acc.w);
    };
    auto lda = [&](int Q) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && Q < nquad) r = gload4(qptr[Q] + row * qstr[Q]);
        return r;
    };
#ifndef QT_SKINNY_INFLIGHT
#define QT_SKINNY_INFLIGHT 4      // (8: +0.02 ms per step, 16: +0.07)
#endif
    for (int Q = 0; Q < nquad; Q += QT_SKINNY_INFLIGHT) {            // A quads in flight per trip (a trip is one dependent memory phase)
        float4 aq[QT_SKINNY_INFLIGHT];
#pragma unroll
        for (int u = 0; u < QT_SKINNY_INFLIGHT; ++u) aq[u] = lda(Q + u);
#pragma unroll
        for (int u = 0; u < QT_SKINNY_INFLIGHT; ++u)
            if (Q + u < nquad) step(aq[u], Q + u);
    }
    const bool post = RPB == 64 && g.post_W != nullptr;       // (uniform; the host guarantees NB == 16: all four waves are here)
    if (!ok && !post) return;
    float4 v = acc;
    if (g.act == QT_ACT_RELU) {
        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
    }
    if (g.act == QT_ACT_RELU_BWD && ok) {          // G = v * relu'(Y): k_act_bwd's arithmetic, Y = g.res (M, res_stride)
        const float4 y = *reinterpret_cast<const float4*>(g.res + row * g.res_stride + j);
        v.x = y.x > 0.0f ? v.x : 0.0f; v.y = y.y > 0.0f ? v.y : 0.0f; v.z = y.z > 0.0f ? v.z : 0.0f; v.w = y.w > 0.0f ? v.w : 0.0f;
    }
    if constexpr (RPB == 64) {
        if (post) {
            // the second product, by wave 0 from the tile in LDS: the same fused multiply-adds in the same order as a k_gemm_skinny<256>
            // launch on the stored rows (quads 0 .. 3 of the row, then the bias quad (1, 0, 0, 0))
            float* h = &hs[(t & 63) * 17 + j];
            h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
            __syncthreads();
            if (t < 64 && ok) {
                float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
                const float* r = &hs[t * 17];
#pragma unroll
                for (int Q = 0; Q <= 4; ++Q) {
                    const float4 a = Q < 4 ? make_float4(r[4 * Q], r[4 * Q + 1], r[4 * Q + 2], r[4 * Q + 3]) : make_float4(1.f, 0.f, 0.f, 0.f);
                    const float4 w0 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 0) * 4);
                    const float4 w1 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 1) * 4);
                    const float4 w2 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 2) * 4);
                    const float4 w3 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 3) * 4);
                    u.x = fmaf(a.x, w0.x, u.x); u.y = fmaf(a.x, w0.y, u.y); u.z = fmaf(a.x, w0.z, u.z); u.w = fmaf(a.x, w0.w, u.w);
                    u.x = fmaf(a.y, w1.x, u.x); u.y = fmaf(a.y, w1.y, u.y); u.z = fmaf(a.y, w1.z, u.z); u.w = fmaf(a.y, w1.w, u.w);
                    u.x = fmaf(a.z, w2.x, u.x); u.y = fmaf(a.z, w2.y, u.y); u.z = fmaf(a.z, w2.z, u.z); u.w = fmaf(a.z, w2.w, u.w);
                    u.x = fmaf(a.w, w3.x, u.x); u.y = fmaf(a.w, w3.y, u.y); u.z = fmaf(a.w, w3.z, u.z); u.w = fmaf(a.w, w3.w, u.w);
                }
                *reinterpret_cast<float4*>(g.post_out + row * 4) = u;
            }
            if (!ok) return;
        }
    }
    if (g.act == QT_ACT_TANH_RES) {
        const float d = g.drop ? g.drop[row] : 1.0f, rs = g.res[row * g.res_stride];
        v.x = tanhf(d * v.x) + rs; v.y = tanhf(d * v.y) + rs; v.z = tanhf(d * v.z) + rs; v.w = tanhf(d * v.w) + rs;
    }
    const int ct = g.Cb + g.Cbb;
    const int pl = j / ct, ch = j - pl * ct;
    if (ch < g.Cb)
        *reinterpret_cast<float4*>(plane_piece(g.out, pl, row, ch, g.Cb, g.M, g.out_sm, g.Cb)) = v;
    else
        *reinterpret_cast<float4*>(plane_piece(g.outb, pl, row, ch - g.Cb, g.Cbb, g.M, g.out_sm, g.Cbb)) = v;
}

// ---- one lane = one node row, ALL 16 output columns: the decoder head's products (fc_out1: 20 -> 16 channels over K = 3 planes,
// model/seq2seq.py:115-121,164-171).  k_gemm_skinny<64> gives a wave 4 of the 16 columns, so the four waves of a workgroup load
// the same 64 operand rows four times over (16 quads each) in four dependent trips; here a lane keeps its row's 16 accumulators,
// every operand quad is loaded ONCE and all of them are in flight together.  W rows come through scalar loads (uniform
// addresses).  The same chain of fused multiply-adds per output element (k ascending): bit-identical to k_gemm_skinny.
// Epilogue as k_gemm_skinny<64>: ReLU / ReLU-backward mask, the second product post_out = [act(out) | 1 0 0 0] @ post_W.
#ifndef QT_ROW16_INF
#define QT_ROW16_INF 8
#endif
__global__ __launch_bounds__(64) void k_gemm_row16(GemmArgs g) {
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    const int t = threadIdx.x;
    const int64_t rows = qt_rows(g.n_dev, g.M);
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    if (row0 >= rows) return;
    const int nquad = g.K >> 2;
    build_quad_table(g.A, qptr, qstr, nquad);          // (nquad <= 64 = the threads of this workgroup: one entry each)
    __syncthreads();
    const int64_t row = row0 + t;
    const bool ok = row < rows;
    float4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto step = [&](float a, int k) {
        const float* __restrict__ wr = g.B + (int64_t)k * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 w = *reinterpret_cast<const float4*>(wr + 4 * c);
            acc[c].x = fmaf(a, w.x, acc[c].x); acc[c].y = fmaf(a, w.y, acc[c].y);
            acc[c].z = fmaf(a, w.z, acc[c].z); acc[c].w = fmaf(a, w.w, acc[c].w);
        }
    };
    constexpr int INF = QT_ROW16_INF;                    // operand quads in flight per trip
    for (int Q = 0; Q < nquad; Q += INF) {
        float4 aq[INF];
#pragma unroll
        for (int u = 0; u < INF; ++u) {
            aq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && Q + u < nquad) aq[u] = gload4(qptr[Q + u] + row * qstr[Q + u]);
        }
#pragma unroll
        for (int u = 0; u < INF; ++u)
            if (Q + u < nquad) {
                step(aq[u].x, 4 * (Q + u)); step(aq[u].y, 4 * (Q + u) + 1); step(aq[u].z, 4 * (Q + u) + 2); step(aq[u].w, 4 * (Q + u) + 3);
            }
    }
    if (!ok) return;
    if (g.act == QT_ACT_RELU) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c].x = fmaxf(acc[c].x, 0.0f); acc[c].y = fmaxf(acc[c].y, 0.0f); acc[c].z = fmaxf(acc[c].z, 0.0f); acc[c].w = fmaxf(acc[c].w, 0.0f);
        }
    }
    if (g.act == QT_ACT_RELU_BWD) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 y = *reinterpret_cast<const float4*>(g.res + row * g.res_stride + 4 * c);
            acc[c].x = y.x > 0.0f ? acc[c].x : 0.0f; acc[c].y = y.y > 0.0f ? acc[c].y : 0.0f;
            acc[c].z = y.z > 0.0f ? acc[c].z : 0.0f; acc[c].w = y.w > 0.0f ? acc[c].w : 0.0f;
        }
    }
    if (g.post_W) {
        float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int Q = 0; Q <= 4; ++Q) {
            const float4 a = Q < 4 ? acc[Q] : make_float4(1.f, 0.f, 0.f, 0.f);
            const float4 w0 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 0) * 4);
            const float4 w1 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 1) * 4);
            const float4 w2 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 2) * 4);
            const float4 w3 = *reinterpret_cast<const float4*>(g.post_W + (4 * Q + 3) * 4);
            u.x = fmaf(a.x, w0.x, u.x); u.y = fmaf(a.x, w0.y, u.y); u.z = fmaf(a.x, w0.z, u.z); u.w = fmaf(a.x, w0.w, u.w);
            u.x = fmaf(a.y, w1.x, u.x); u.y = fmaf(a.y, w1.y, u.y); u.z = fmaf(a.y, w1.z, u.z); u.w = fmaf(a.y, w1.w, u.w);
            u.x = fmaf(a.z, w2.x, u.x); u.y = fmaf(a.z, w2.y, u.y); u.z = fmaf(a.z, w2.z, u.z); u.w = fmaf(a.z, w2.w, u.w);
            u.x = fmaf(a.w, w3.x, u.x); u.y = fmaf(a.w, w3.y, u.y); u.z = fmaf(a.w, w3.z, u.z); u.w = fmaf(a.w, w3.w, u.w);
        }
        *reinterpret_cast<float4*>(g.post_out + row * 4) = u;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<float4*>(g.out + row * 16 + 4 * c) = acc[c];
}

// ---- the decoder head's backward products in ONE launch, one lane per node row:
//   G = relu'(Y) (.) (gU @ Wb2)          (N, 16): the gradient at fc_out1's output (Wb2 (4, 16) = the coefficient columns of
//                                         fc_out2 transposed); stored -- the deferred weight gradient of fc_out1 reads it
//   planes = G @ Wb1^T                   (K, N, Ca | Cbb): the data gradient of fc_out1, Wb1 (K (Ca + Cbb), 16) = its rows
// Replaces a k_gemm_skinny launch (gU -> G) and a k_gemm_fwd<2, 128> launch (G -> planes: 16 x 60 on the MFMA) whose operand G made
// a round trip through memory in between: 25 -> ~11 us per decoder step.  The chains are the k-ordered fused multiply-adds of the
// two launches (v_mfma_f32_32x32x2_f32 accumulates in k order): bit-identical planes.
struct HeadDgradArgs {
    const float *gU, *Wb2, *Y, *Wb1;
    float *G, *out, *outb;
    int N, K, Cb, Cbb, out_sm;
    const int32_t* n_dev;
};
__global__ __launch_bounds__(64) void k_head_dgrad(HeadDgradArgs g) {
    const int64_t rows = qt_rows(g.n_dev, g.N);
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if ((int64_t)blockIdx.x * 64 >= rows) return;
    const bool ok = row < rows;
    const int64_t r = ok ? row : rows - 1;              // (clamped loads, predicated stores)
    const float4 gu = gload4(g.gU + r * 4);
    float4 y[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = gload4(g.Y + r * 16 + 4 * c);
    float G[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        const float guv[4] = {gu.x, gu.y, gu.z, gu.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 w = *reinterpret_cast<const float4*>(g.Wb2 + k * 16 + 4 * c);
            a.x = fmaf(guv[k], w.x, a.x); a.y = fmaf(guv[k], w.y, a.y); a.z = fmaf(guv[k], w.z, a.z); a.w = fmaf(guv[k], w.w, a.w);
        }
        a.x = y[c].x > 0.0f ? a.x : 0.0f; a.y = y[c].y > 0.0f ? a.y : 0.0f; a.z = y[c].z > 0.0f ? a.z : 0.0f; a.w = y[c].w > 0.0f ? a.w : 0.0f;
        G[4 * c] = a.x; G[4 * c + 1] = a.y; G[4 * c + 2] = a.z; G[4 * c + 3] = a.w;
        if (ok) *reinterpret_cast<float4*>(g.G + row * 16 + 4 * c) = a;
    }
    const int ct = g.Cb + g.Cbb;
    for (int pl = 0; pl < g.K; ++pl) {
        for (int ch = 0; ch < ct; ch += 4) {
            const float* __restrict__ wr = g.Wb1 + (int64_t)(pl * ct + ch) * 16;      // rows of the four output columns (uniform)
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // the reduction order of k_gemm_fwd's MFMA stream (v_mfma_f32_32x32x2_f32: lanes 0-31 carry k = 8 j + i, lanes 32-63
                // k = 8 j + 4 + i of instruction i): 0 4 1 5 2 6 3 7 | 8 12 9 13 10 14 11 15 -- bit-identical planes
                float a = 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 wl = *reinterpret_cast<const float4*>(wr + q * 16 + 8 * j);
                    const float4 wh = *reinterpret_cast<const float4*>(wr + q * 16 + 8 * j + 4);
                    a = fmaf(G[8 * j + 0], wl.x, a); a = fmaf(G[8 * j + 4], wh.x, a);
                    a = fmaf(G[8 * j + 1], wl.y, a); a = fmaf(G[8 * j + 5], wh.y, a);
                    a = fmaf(G[8 * j + 2], wl.z, a); a = fmaf(G[8 * j + 6], wh.z, a);
                    a = fmaf(G[8 * j + 3], wl.w, a); a = fmaf(G[8 * j + 7], wh.w, a);
                }
                o[q] = a;
            }
            if (!ok) continue;
            const float4 v = make_float4(o[0], o[1], o[2], o[3]);
            if (ch < g.Cb)
                *reinterpret_cast<float4*>(plane_piece(g.out, pl, row, ch, g.Cb, g.N, g.out_sm, g.Cb)) = v;
            else
                *reinterpret_cast<float4*>(plane_piece(g.outb, pl, row, ch - g.Cb, g.Cbb, g.N, g.out_sm, g.Cbb)) = v;
        }
    }
}

// ---- bf16x3 variant of the forward / data-gradient GEMM -------------------------------------------------------------
// fp32 MFMA runs at the VALU FLOP rate on gfx950 and bounds k_gemm_fwd (HISTORY.md section C).  Here every fp32 operand
// is split into three bf16 terms (x = hi + mid + lo, each rounded to nearest) and a product group is six bf16 MFMAs
// (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) accumulated in fp32: the dropped terms are O(2^-24) of the product,
// i.e. fp32-level error, at 16 k per 32-cycle instruction instead of 2 k per 64-cycle instruction.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16* h, __bf16* m, __bf16* l) {
    const __bf16 hh = (__bf16)x;
    const float r = x - (float)hh;
    const __bf16 mm = (__bf16)r;
    const float r2 = r - (float)mm;
    *h = hh; *m = mm; *l = (__bf16)r2;
}

template <int NT, int KWT>
__global__ __launch_bounds__(256) void k_gemm_fwd3(GemmArgs g) {
    constexpr int BNT = 32 * NT;
    constexpr int PITCH = KWT + 8;          // bf16 elements; row pitch in bytes = 16 (mod 32): conflict-free ds_read_b128
    constexpr int PLANE = BNT * PITCH;      // one of the three split planes of W^T: Bt[x][column][k]
    static_assert(3 * PLANE * 2 >= BM * 64 * 4, "LDS staging tile does not fit in the W buffer");
    __shared__ __attribute__((aligned(16))) __bf16 Bt[3 * PLANE];
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * BM;
    const int j0 = blockIdx.y * BNT;
    const int64_t rows = qt_rows(g.n_dev, g.M);
    if (i0 >= rows) return;
    const int nquad = g.K >> 2;
    build_quad_table(g.A, qptr, qstr, nquad);
    const int64_t my_row = i0 + wave * 32 + l32;
    const bool row_ok = my_row < rows;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    for (int k0 = 0; k0 < g.K; k0 += KWT) {
        const int kn = min(KWT, g.K - k0);           // multiple of 4
        const int kn16 = (kn + 15) & ~15;
        __syncthreads();
        // W chunk -> LDS, transposed and split: pairs of k rows so that every LDS store is a packed 32-bit word
        for (int e = t; e < (kn16 / 2) * (BNT / 4); e += 256) {
            const int kp = e / (BNT / 4), jq = (e % (BNT / 4)) * 4;
            float w[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kb = 2 * kp + u;
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kb < kn && j0 + jq < g.NB) f = *reinterpret_cast<const float4*>(g.B + (int64_t)(k0 + kb) * g.NB + j0 + jq);
                w[u][0] = f.x; w[u][1] = f.y; w[u][2] = f.z; w[u][3] = f.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                __bf16 h0, m0, l0, h1, m1, l1;
                split3(w[0][c], &h0, &m0, &l0);
                split3(w[1][c], &h1, &m1, &l1);
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                const int o = (jq + c) * PITCH + 2 * kp;
                *reinterpret_cast<bf16x2*>(&Bt[o]) = bf16x2{h0, h1};
                *reinterpret_cast<bf16x2*>(&Bt[PLANE + o]) = bf16x2{m0, m1};
                *reinterpret_cast<bf16x2*>(&Bt[2 * PLANE + o]) = bf16x2{l0, l1};
            }
        }
        __syncthreads();
        // MFMA stream: step J covers k = 16 J .. 16 J + 15; this lane feeds k = 16 J + 8 half .. + 7 (two quads of its row)
        const int q0 = k0 >> 2, qn = kn >> 2, nJ = kn16 >> 4;
        auto ldq = [&](int q) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && q < qn) r = gload4(qptr[q0 + q] + my_row * qstr[q0 + q]);
            return r;
        };
        float4 c0 = ldq(2 * half), c1 = ldq(2 * half + 1);                 // J = 0
        float4 n0 = ldq(4 + 2 * half), n1 = ldq(4 + 2 * half + 1);         // J = 1
        for (int J = 0; J < nJ; ++J) {
            const float av[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            c0 = n0; c1 = n1;
            n0 = ldq(4 * (J + 2) + 2 * half);
            n1 = ldq(4 * (J + 2) + 2 * half + 1);
            bf16x8 ah, am, al;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __bf16 h, m, l;
                split3(av[i], &h, &m, &l);
                ah[i] = h; am[i] = m; al[i] = l;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int o = (nt * 32 + l32) * PITCH + 16 * J + 8 * half;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bt[o]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Bt[PLANE + o]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bt[2 * PLANE + o]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);     // small terms first
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
            }
        }
    }
    float* Cs = reinterpret_cast<float*>(Bt);    // 128 rows x 64 columns per pass
#pragma unroll
    for (int h2 = 0; h2 < NT / 2; ++h2) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nt = 2 * h2 + u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + u * 32 + l32] = acc[nt][r];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BM * 16 / 256; ++u) {
            const int e = t + 256 * u;
            const int row = e >> 4, c4 = (e & 15) * 4;
            const int64_t i = i0 + row;
            const int j = j0 + h2 * 64 + c4;
            if (i >= rows || j >= g.NB) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * 64 + c4]);
            if (g.act == QT_ACT_RELU) {
                v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
            }
            if (g.act == QT_ACT_TANH_RES) {
                const float d = g.drop ? g.drop[i] : 1.0f, rs = g.res[i * g.res_stride];
                v.x = tanhf(d * v.x) + rs; v.y = tanhf(d * v.y) + rs; v.z = tanhf(d * v.z) + rs; v.w = tanhf(d * v.w) + rs;
            }
            const int ct = g.Cb + g.Cbb;
            const int pl = j / ct, ch = j - pl * ct;
            if (ch < g.Cb)
                *reinterpret_cast<float4*>(plane_piece(g.out, pl, i, ch, g.Cb, g.M, g.out_sm, g.Cb)) = v;
            else
                *reinterpret_cast<float4*>(plane_piece(g.outb, pl, i, ch - g.Cb, g.Cbb, g.M, g.out_sm, g.Cbb)) = v;
        }
    }
}

// Data-gradient GEMM as a split-bf16 product (gradients only): out planes = A (N x K fp32 rows) @ B, with B^T given as two bf16
// terms per element (Bhi + Blo ~ B, qt_split_bf16) and A split on the fly; three bf16 MFMAs per product group (lo.hi, hi.lo, hi.hi:
// the dropped lo.lo term is 2^-16 of the product) at 16 k per 32-cycle instruction.  The fp32-MFMA form of this product runs at
// 60 % of the fp32-MFMA peak for hidden 32 (K = 128): the matrix pipe, not memory, sets its time.
template <int NT>
__global__ __launch_bounds__(256, 4) void k_gemm_sb(GemmArgs g, const __bf16* __restrict__ Bhi, const __bf16* __restrict__ Blo) {
    constexpr int KWT = 64, BNT = 32 * NT;
    constexpr int PITCH = KWT + 8;          // bf16 elements; row pitch in bytes = 16 (mod 32): conflict-free ds_read_b128
    constexpr int PLANE = BNT * PITCH;
    constexpr int LDS_BYTES = 2 * PLANE * 2 > BM * 64 * 4 ? 2 * PLANE * 2 : BM * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    __bf16* Bt = reinterpret_cast<__bf16*>(lds);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * BM;
    const int j0 = blockIdx.y * BNT;
    const int64_t rows = qt_rows(g.n_dev, g.M);
    if (i0 >= rows) return;
    const int64_t my_row = i0 + wave * 32 + l32;
    const bool row_ok = my_row < rows;
    const float* arow = g.A.a0 + my_row * g.A.lda0;
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (int k0 = 0; k0 < g.K; k0 += KWT) {
        const int kn = min(KWT, g.K - k0);           // multiple of 16 (checked by the host entry)
        __syncthreads();
        const int k8 = kn >> 3;
        for (int e = t; e < BNT * k8; e += 256) {     // both bf16 planes of the weight chunk: straight 16-byte copies
            const int c = e / k8, kq = e - c * k8;
            u32x4 wh = {0u, 0u, 0u, 0u}, wl = wh;
            if (j0 + c < g.NB) {
                const int64_t o = (int64_t)(j0 + c) * g.K + k0 + 8 * kq;
                wh = *reinterpret_cast<const u32x4*>(Bhi + o);
                wl = *reinterpret_cast<const u32x4*>(Blo + o);
            }
            *reinterpret_cast<u32x4*>(&Bt[c * PITCH + 8 * kq]) = wh;
            *reinterpret_cast<u32x4*>(&Bt[PLANE + c * PITCH + 8 * kq]) = wl;
        }
        __syncthreads();
        const int nJ = kn >> 4;
        auto ldk = [&](int J, int u) {              // quad u of this lane's eight k of step J
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok && J < nJ) r = gload4(arow + k0 + 16 * J + 8 * half + 4 * u);
            return r;
        };
        float4 c0 = ldk(0, 0), c1 = ldk(0, 1), n0 = ldk(1, 0), n1 = ldk(1, 1);
        for (int J = 0; J < nJ; ++J) {
            const float av[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            c0 = n0; c1 = n1;
            n0 = ldk(J + 2, 0);
            n1 = ldk(J + 2, 1);
            bf16x8 ah, al;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const __bf16 h = (__bf16)av[i];
                ah[i] = h;
                al[i] = (__bf16)(av[i] - (float)h);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int o = (nt * 32 + l32) * PITCH + 16 * J + 8 * half;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bt[o]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bt[PLANE + o]);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);     // small terms first
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
            }
        }
    }
    float* Cs = reinterpret_cast<float*>(lds);    // 128 rows x 64 columns per pass
#pragma unroll
    for (int h2 = 0; h2 < NT / 2; ++h2) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nt = 2 * h2 + u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + u * 32 + l32] = acc[nt][r];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < BM * 16 / 256; ++u) {
            const int e = t + 256 * u;
            const int row = e >> 4, c4 = (e & 15) * 4;
            const int64_t i = i0 + row;
            const int j = j0 + h2 * 64 + c4;
            if (i >= rows || j >= g.NB) continue;
            const float4 v = *reinterpret_cast<const float4*>(&Cs[row * 64 + c4]);
            const int ct = g.Cb + g.Cbb;
            const int pl = j / ct, ch = j - pl * ct;
            if (ch < g.Cb)
                *reinterpret_cast<float4*>(plane_piece(g.out, pl, i, ch, g.Cb, g.M, g.out_sm, g.Cb)) = v;
            else
                *reinterpret_cast<float4*>(plane_piece(g.outb, pl, i, ch - g.Cb, g.Cbb, g.M, g.out_sm, g.Cbb)) = v;
        }
    }
}

constexpr int WR = 32;
// FW feature waves x (4 / FW) row groups, CT column tiles of 32.  The MFMA count per pass is what bounds this kernel, so
// a narrow weight must not pay for the full 128 x 64 tile: with FW < 4 the spare waves take a share of every pass's rows
// (their partial tiles are added through LDS at the end, in a fixed order), with CT = 1 the second column tile is skipped.
template <int FW, int CT>
__device__ __forceinline__ void wgrad_body(const PlaneSrc& A, const float* __restrict__ G, int M, int NB, int64_t rbeg,
                                           int64_t rend, float* obase, int accumulate, int jt, int ldg, int gpl = 0,
                                           int64_t gps = 0) {       // gpl > 0: G columns in planes of gpl floats, gps apart
    constexpr int RG = 4 / FW;             // row groups
    constexpr int KS = WR / 2 / RG;        // k-steps (2 rows each) per row group and pass
    constexpr int BMF = FW * 32;           // features per block
    // A tiles hold only the BMF features this block owns (narrow weights then fit 4 workgroups per CU); the same floats
    // later park the row groups' partial tiles
    constexpr int RED_FLOATS = (RG - 1) * FW * CT * 16 * 64;
    constexpr int AS_FLOATS = 2 * WR * BMF > RED_FLOATS ? 2 * WR * BMF : RED_FLOATS;
    __shared__ __attribute__((aligned(16))) float As_pool[AS_FLOATS];
    auto As = [&](int buf, int row, int col) -> float& { return As_pool[(buf * WR + row) * BMF + col]; };
    __shared__ __attribute__((aligned(16))) float Gs[2][WR][BN];
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const int fw = wave % FW, rg = wave / FW;
    const int f0 = blockIdx.x * BMF, j0 = jt * BN;
    const int nquad = M >> 2;
    build_quad_table(A, qptr, qstr, nquad);
    __syncthreads();
    // staging roles: A tile = 32 rows x 32 quads -> 4 float4 per thread; G tile = 32 rows x 16 quads -> 2 per thread.
    // Within one load instruction the 8 threads of a row take 8 CONSECUTIVE quads (128 contiguous bytes where the quads
    // share a plane part); giving each thread 4 consecutive quads instead made every 4-lane group of the texture addresser
    // span four 64-byte segments.
    const int a_row = t >> 3, a_q = t & 7;
    const int g_row = t >> 3, g_q = t & 7;
    // two passes of operands in flight: the rows of this kernel come from HBM (activations saved by the forward pass),
    // and with one pass of prefetch every group of weights ran at ~3 TB/s whatever its MFMA load
    float4 pa[2][4], pg[2][2];
    auto fetch = [&](int64_t r0, float4 (&qa)[4], float4 (&qg)[2]) {
        const int64_t ra = r0 + a_row;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ql = u * 8 + a_q;
            const int Q = (f0 >> 2) + ql;
            qa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ra < rend && Q < nquad && ql < BMF / 4) qa[u] = gload4(qptr[Q] + ra * qstr[Q]);
        }
        const int64_t rgw = r0 + g_row;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int jq = (u * 8 + g_q) * 4;
            qg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rgw < rend && j0 + jq < NB && jq < CT * 32) {
                const int j = j0 + jq, pl = gpl ? j / gpl : 0;
                qg[u] = gload4(G + pl * gps + rgw * ldg + (j - pl * gpl));
            }
        }
    };
    auto stash = [&](int buf, const float4 (&qa)[4], const float4 (&qg)[2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((u * 8 + a_q) * 4 < BMF) *reinterpret_cast<float4*>(&As(buf, a_row, (u * 8 + a_q) * 4)) = qa[u];
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<float4*>(&Gs[buf][g_row][(u * 8 + g_q) * 4]) = qg[u];
    };
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    auto mfma_pass = [&](int buf) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int ks = rg * KS + k;
            const float a = As(buf, 2 * ks + half, fw * 32 + l32);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, Gs[buf][2 * ks + half][c * 32 + l32], acc[c], 0, 0, 0);
        }
    };
    if (rbeg < rend) {
        // LDS buffer b holds pass p (p even: b = 0), register set (p + 1) & 1 holds pass p + 1, set p & 1 is loaded with
        // pass p + 2 while pass p is multiplied; rows past `rend` load as zeros, so the tail needs no special case
        fetch(rbeg, pa[0], pg[0]);
        fetch(rbeg + WR, pa[1], pg[1]);
        stash(0, pa[0], pg[0]);
        __syncthreads();
        for (int64_t r0 = rbeg; r0 < rend; r0 += 2 * WR) {
            fetch(r0 + 2 * WR, pa[0], pg[0]);
            mfma_pass(0);
            stash(1, pa[1], pg[1]);
            __syncthreads();
            fetch(r0 + 3 * WR, pa[1], pg[1]);
            mfma_pass(1);                        // an all-zero pass when r0 + WR >= rend
            stash(0, pa[0], pg[0]);
            __syncthreads();
        }
    }
    if constexpr (RG > 1) {
        // add the row groups' partial tiles: groups 1.. park theirs in LDS (the A buffers are free), group 0 adds in order
        float* red = As_pool;
        if (rg > 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((((rg - 1) * FW + fw) * CT + c) * 16 + r) * 64 + lane] = acc[c][r];
        }
        __syncthreads();
        if (rg > 0) return;
#pragma unroll
        for (int g2 = 1; g2 < RG; ++g2)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] += red[((((g2 - 1) * FW + fw) * CT + c) * 16 + r) * 64 + lane];
    }
#pragma unroll
    for (int jt = 0; jt < CT; ++jt) {
        const int j = j0 + jt * 32 + l32;
        if (j >= NB) continue;
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {      // all slab reads first: 16 independent loads in flight
            const int i = f0 + fw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            old[r] = (accumulate && i < M) ? obase[(int64_t)i * NB + j] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = f0 + fw * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < M) obase[(int64_t)i * NB + j] = old[r] + acc[jt][r];
        }
    }
}

// MODE 1: part[blockIdx.z] = A[rows]^T @ G[rows] over this block's row chunk.  Block = 32 FW features x 32 CT
// columns of the weight gradient; the reduction runs over node rows in passes of 32 rows: float4 global loads ->
// registers (prefetch of the next pass) -> double-buffered LDS.
template <int FW, int CT>
__global__ __launch_bounds__(256) void k_gemm_wgrad(GemmArgs g) {
    const int64_t rbeg = (int64_t)blockIdx.z * g.row0_step;
    const int64_t rend = min((int64_t)qt_rows(g.n_dev, g.A.N), rbeg + g.row0_step);
    wgrad_body<FW, CT>(g.A, g.B, g.M, g.NB, rbeg, rend, g.out + (int64_t)blockIdx.z * g.M * g.NB, g.accumulate, blockIdx.y, g.NB);
}

// The same reduction for up to 16 uses of ONE weight in a single launch (the rollout steps of a pass): z-blocks
// [zend[s-1], zend[s]) walk the node rows of use s; every z-block owns its slab, qt_colsum adds them in fixed order.
constexpr int MAXSEG = 16;
struct WgradGroup {
    const float* a0[MAXSEG];
    const float* a_rest[MAXSEG];
    const float* a0b[MAXSEG];
    const float* a_restb[MAXSEG];
    const float* S[MAXSEG];
    const float* G[MAXSEG];
    const int32_t* n_dev[MAXSEG];
    int N[MAXSEG], zend[MAXSEG], lda0[MAXSEG], lda0b[MAXSEG];
    int nseg, Ka, Ca, Cab, Ks, Co, rows, sm;
    float* part;
    // Gn weights per use (the stacks of one layer, qt_proj_group): grid y = (group, column tile); group g reads plane 0 at
    // a0 + g gsA and the gradient rows at G + g gsG (row stride ldg), and owns slab (z, g) of part
    int Gn, ytiles, ldg, gpl, per_node;      // per_node: gsA / gsG are floats per node of the use (x N[s])
    int64_t gsA, gsG;
};
template <int FW, int CT>
__global__ __launch_bounds__(256) void k_gemm_wgrad_group(WgradGroup w) {
    int s = 0;
    while (s + 1 < w.nseg && (int)blockIdx.z >= w.zend[s]) ++s;
    const int zl = blockIdx.z - (s ? w.zend[s - 1] : 0);
    PlaneSrc A;
    A.a0 = w.a0[s]; A.a_rest = w.a_rest[s]; A.a0b = w.a0b[s]; A.a_restb = w.a_restb[s]; A.S = w.S[s];
    A.Ka = w.Ka; A.Ca = w.Ca; A.Cab = w.Cab; A.Ks = w.Ks; A.N = w.N[s]; A.lda0 = w.lda0[s]; A.lda0b = w.lda0b[s]; A.sm = w.sm;
    const int M = w.Ka * (w.Ca + w.Cab) + w.Ks;
    const int64_t rbeg = (int64_t)zl * w.rows;
    const int64_t rend = min((int64_t)qt_rows(w.n_dev[s], w.N[s]), rbeg + w.rows);
    const int grp = blockIdx.y / w.ytiles, jt = blockIdx.y - grp * w.ytiles;
    const int64_t gmul = w.per_node ? (int64_t)grp * w.N[s] : grp;
    A.a0 += gmul * w.gsA;
    wgrad_body<FW, CT>(A, w.G[s] + gmul * w.gsG, M, w.Co, rbeg, rend, w.part + ((int64_t)blockIdx.z * w.Gn + grp) * M * w.Co, 0, jt, w.ldg, w.gpl,
                       (int64_t)w.N[s] * w.gpl);
}

// 32 columns x 32 row groups per workgroup, four slabs in flight per thread; fixed summation order (deterministic).
// (With 8 row groups and one load in flight the 640 slabs of a grouped weight gradient took 30 us: a latency chain.)
__global__ __launch_bounds__(1024) void k_colsum(const float* __restrict__ part, int nblk, int64_t len, float* __restrict__ out) {
    __shared__ float sm[32][33];
    const int cl = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int64_t c = (int64_t)blockIdx.x * 32 + cl;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (c < len) {
        int i = r;
        for (; i + 96 < nblk; i += 128) {
            a0 += part[(int64_t)i * len + c];
            a1 += part[(int64_t)(i + 32) * len + c];
            a2 += part[(int64_t)(i + 64) * len + c];
            a3 += part[(int64_t)(i + 96) * len + c];
        }
        for (; i < nblk; i += 32) a0 += part[(int64_t)i * len + c];
    }
    sm[r][cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (r == 0 && c < len) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += sm[k][cl];
        out[c] = s;
    }
}

// tile variant by weight shape: feature waves 1 / 2 / 4 for up to 32 / 64 / more features, one column tile up to 32 columns
inline int wgrad_fw(int M) { return M <= 32 ? 1 : (M <= 64 ? 2 : 4); }
#define QT_WGRAD_DISPATCH(K, M_, Co_, grid_, stream_, arg_)                                                         \
    do {                                                                                                            \
        const int fw_ = wgrad_fw(M_);                                                                               \
        const bool one_ = (Co_) <= 32;                                                                              \
        if (fw_ == 1 && one_) hipLaunchKernelGGL((K<1, 1>), grid_, dim3(256), 0, (hipStream_t)(stream_), arg_);     \
        else if (fw_ == 1) hipLaunchKernelGGL((K<1, 2>), grid_, dim3(256), 0, (hipStream_t)(stream_), arg_);        \
        else if (fw_ == 2 && one_) hipLaunchKernelGGL((K<2, 1>), grid_, dim3(256), 0, (hipStream_t)(stream_), arg_); \
        else if (fw_ == 2) hipLaunchKernelGGL((K<2, 2>), grid_, dim3(256), 0, (hipStream_t)(stream_), arg_);        \
        else if (one_) hipLaunchKernelGGL((K<4, 1>), grid_, dim3(256), 0, (hipStream_t)(stream_), arg_);            \
        else hipLaunchKernelGGL((K<4, 2>), grid_, dim3(256), 0, (hipStream_t)(stream_), arg_);                      \
    } while (0)
constexpr int WGRAD_ROWS = 512;
#ifndef QT_WG_ROWS
#define QT_WG_ROWS 512      // 11.02 ms per training step against 11.07 (1024) and 11.20 (256)
#endif
constexpr int WGRAD_GROUP_ROWS = QT_WG_ROWS;


// ---- cell backward fused into the data-gradient GEMM of the gate weights (hidden 8 / 16).
// gG = d loss / d gate pre-activations comes out of the cell backward (k_lstm_bwd's arithmetic, qt_cell.h) and is at once
// the left operand of  gT = gG W^T  (K = 4h reduction, all output planes in this workgroup's 32 NT columns).  Here a
// workgroup computes the gG rows of its 128 nodes into LDS (and to global memory: the deferred weight gradient reads them),
// then feeds the MFMA loop from LDS: the (N, 4h) matrix is not read back from memory and one launch per use is gone.
// Same operand order as k_gemm_fwd on the stored gG: bit-identical planes.
struct DgradCellArgs {
    const float *gO, *gHn, *gCn, *gates, *Cprev, *wc, *ln;
    int ld_go, ld_gh, ld_gc, ld_c, h;
    float *gG, *gCprev, *part;
    int accumulate;
    const float* BT;        // (NB, 4h): rows k*C + c of the forward weight (= the transposed right operand)
    const __bf16 *BThi, *BTlo;   // optional: the same rows split into two bf16 terms (qt_split_bf16): the product runs on bf16 MFMA
    int M, NB, Kb, Cb, Cbb;
    float *out, *outb;
    const int32_t* n_dev;
    int out_sm;             // output planes 1 .. Kb-1 slice-major (plane_piece)
    // Two gradient sums that autograd would otherwise make with separate elementwise launches (a tensor with two consumers):
    const float* gHn2;      // optional second gradient of H' (the state goes to the next time step AND to the next layer): added on load
    int ld_gh2;
    const float* add0;      // optional (N, Cb): added to output plane 0 of part a (the decoder input is also the head's residual
                            // operand: that gradient rides into the Clenshaw recurrence as part of A_0)
};

// BG: the right operand (the weight rows, <= 32 KB, L1 / L2 resident) is read straight from global memory by the lanes that
// need it instead of being staged in LDS: the workgroup's LDS drops from 64 KB to 38 KB, so THREE workgroups fit a CU instead
// of two -- more workgroups whose load / arithmetic / MFMA / store phases overlap.
#ifndef QT_DGRAD_BG
#define QT_DGRAD_BG 1
#endif
#ifndef QT_DGRAD_OCC
#define QT_DGRAD_OCC 4      // 128 VGPRs (8-10 spilled): FOUR workgroups per CU = all 940 tiles of the bench shape resident at once
#endif                    // (3: 138-155 VGPRs, 768 resident + a second round; 8.41 -> 8.34 ms per frozen step)
template <int NT, int LPN, bool BG = (QT_DGRAD_BG != 0)>
__global__ __launch_bounds__(256, BG ? QT_DGRAD_OCC : 2) void k_dgrad_cell(DgradCellArgs g) {
    using namespace qtcell;
    constexpr int BNT = 32 * NT, K = 16 * LPN, PITCH = K + 4, RP = 256 / LPN;
    __shared__ __attribute__((aligned(16))) float Bt[BG ? 4 : BNT * PITCH];
    __shared__ __attribute__((aligned(16))) float As[128 * (PITCH > 68 ? PITCH : 68)];       // (>= 128 x 68: the epilogue's staging tile)
    __shared__ float sm[4 * LPN * 11 * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * BM;
    const int64_t rows = qt_rows(g.n_dev, g.M);
    const int h = g.h;
    if (i0 >= rows) return;        // past the valid rows: nothing to add to the partials (the slab rows start at zero)
    // W chunk (all of it: K = 4h fits one pass) -> LDS; independent of the cell phase below
    if constexpr (!BG) {
        for (int e = t; e < BNT * (K / 4); e += 256) {
            const int c = e / (K / 4), kq = e - c * (K / 4);
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < g.NB) w = *reinterpret_cast<const float4*>(g.BT + (int64_t)c * K + 4 * kq);
            *reinterpret_cast<float4*>(&Bt[c * PITCH + 4 * kq]) = w;
        }
    }
    // cell backward of this workgroup's rows (rows past the valid count contribute zeros)
    {
        const int j0 = (t % LPN) * 4;
        const F4 wci = ld4(g.wc + j0), wcf = ld4(g.wc + h + j0), wco = ld4(g.wc + 2 * h + j0);
        F4 gam_h = {{1, 1, 1, 1}}, gam_c = {{1, 1, 1, 1}};
        if (g.ln) {
            gam_h = ld4(g.ln + j0);
            gam_c = ld4(g.ln + 2 * h + j0);
        }
        float acc[11][4];
#pragma unroll
        for (int a = 0; a < 11; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[a][k] = 0.0f;
#pragma unroll
        for (int r0 = 0; r0 < BM; r0 += RP) {
            const int r = r0 + t / LPN;
            const int64_t node = i0 + r;
            const bool ok = node < rows;
            const F4 z = {{0, 0, 0, 0}};
            F4 I = z, F = z, T = z, Og = z, cp = z, gyh = z, gyc = z, go_in = z;
            if (ok) {
                const float* gs = g.gates + node * 4 * h + j0;
                I = ld4(gs); F = ld4(gs + h); T = ld4(gs + 2 * h); Og = ld4(gs + 3 * h);
                if (g.Cprev) cp = ld4(g.Cprev + node * g.ld_c + j0);
                if (g.gHn) gyh = ld4(g.gHn + node * g.ld_gh + j0);
                if (g.gCn) gyc = ld4(g.gCn + node * g.ld_gc + j0);
                if (g.gO) go_in = ld4(g.gO + node * g.ld_go + j0);
                if (g.gHn2) {
                    const F4 h2 = ld4(g.gHn2 + node * g.ld_gh2 + j0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) gyh.v[k] += h2.v[k];
                }
            }
#ifdef QT_EXP_DG_NOCELL
            CellBwdOut o;
            o.ggi = I; o.ggf = F; o.ggc = T; o.ggo = Og; o.gcp = cp;
            acc[0][0] += gyh.v[0] + gyc.v[0] + go_in.v[0] + wci.v[0] + wcf.v[0] + wco.v[0] + gam_h.v[0] + gam_c.v[0];
#else
            const CellBwdOut o = cell_backward<LPN>(I, F, T, Og, cp, gyh, gyc, go_in, wci, wcf, wco, gam_h, gam_c,
                                                    g.ln != nullptr, h, acc);
#endif
            float* as = As + r * PITCH + j0;
            st4(as, o.ggi); st4(as + h, o.ggf); st4(as + 2 * h, o.ggc); st4(as + 3 * h, o.ggo);
            if (ok) {
#ifndef QT_EXP_DG_NOGG
                float* gg = g.gG + node * 4 * h + j0;
                st4(gg, o.ggi); st4(gg + h, o.ggf); st4(gg + 2 * h, o.ggc); st4(gg + 3 * h, o.ggo);
#endif
                if (g.gCprev) st4(g.gCprev + node * h + j0, o.gcp);
            }
        }
        block_param_reduce<LPN, 11>(acc, h, sm, g.part + (int64_t)blockIdx.x * 11 * h, g.accumulate);
    }
    qt_lds_barrier();                                 // Bt, As complete
    f32x16 acc2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[nt][r] = 0.0f;
    if (g.BThi) {
        // OPT-IN split-bf16 product (ops.DGRAD_SPLIT_BF16; backward only; the default is the exact fp32 branch below): gG = hi + lo, W = Whi + Wlo (two bf16 terms each, the
        // weight split once per pass by qt_split_bf16), gG W^T ~ hi Whi + hi Wlo + lo Whi -- relative error ~2^-16 per
        // product -- on v_mfma_f32_32x32x16_bf16: 3 MFMAs of 32 cycles per 16 k instead of 8 fp32 MFMAs of 64 cycles (the
        // fp32 MFMA issues on the vector pipe: its 10.7 us per launch at the bench shape added to the cell arithmetic).
        // Lane (r = l & 31, hh = l >> 5) holds A[row r][k = 16 s + 8 hh + j] and B[k = 16 s + 8 hh + j][column r], j = 0 .. 7.
#pragma unroll
        for (int s_ = 0; s_ < K / 16; ++s_) {
            const float* ap = &As[(wave * 32 + l32) * PITCH + 16 * s_ + 8 * half];
            const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
            const float af[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            bf16x8 ahi, alo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const __bf16 hq = (__bf16)af[q];
                ahi[q] = hq;
                alo[q] = (__bf16)(af[q] - (float)hq);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = nt * 32 + l32;                  // (columns past NB: clamped load, zeroed value)
                const int64_t off = (int64_t)(c < g.NB ? c : 0) * K + 16 * s_ + 8 * half;
                bf16x8 bhi = *(const __attribute__((address_space(1))) bf16x8*)(g.BThi + off);
                bf16x8 blo = *(const __attribute__((address_space(1))) bf16x8*)(g.BTlo + off);
                if (c >= g.NB) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { bhi[q] = (__bf16)0.0f; blo[q] = (__bf16)0.0f; }
                }
#ifdef QT_EXP_DG_NOMFMA
                acc2[nt][0] += (float)ahi[0] * (float)bhi[0] + (float)alo[1] * (float)blo[1];
#else
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc2[nt], 0, 0, 0);
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc2[nt], 0, 0, 0);
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc2[nt], 0, 0, 0);
#endif
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < K / 8; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(&As[(wave * 32 + l32) * PITCH + 8 * j + 4 * half]);
        float4 bq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (BG) {
                const int c = nt * 32 + l32;                  // (columns past NB: clamped load, zeroed value)
                const float4 w = gload4(g.BT + (int64_t)(c < g.NB ? c : 0) * K + 8 * j + 4 * half);
                bq[nt] = c < g.NB ? w : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                bq[nt] = *reinterpret_cast<const float4*>(&Bt[(nt * 32 + l32) * PITCH + 8 * j + 4 * half]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[nt].x, acc2[nt], 0, 0, 0);
            acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[nt].y, acc2[nt], 0, 0, 0);
            acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[nt].z, acc2[nt], 0, 0, 0);
            acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[nt].w, acc2[nt], 0, 0, 0);
        }
    }
    }
    // epilogue as in k_gemm_fwd: the tile goes through LDS (As is free now) so that rows leave as float4 pieces.  The staging
    // tile has a pitch of 68 floats (As holds 128 x 68): with slice-major output planes a wave stores 64 consecutive ROWS of one
    // 4-channel piece -- 1 KB contiguous in that slice's array -- and reads them from LDS at a 272-byte stride, which the 64
    // banks take without conflicts (a 256-byte stride would hit one bank group 16 times).
    float* Cs = As;
    constexpr int CP = 68;
#pragma unroll
    for (int h2 = 0; h2 < (NT + 1) / 2; ++h2) {
        qt_lds_barrier();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int nt = 2 * h2 + u;
            if (nt < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + u * 32 + l32] = acc2[nt][r];
            }
        }
        qt_lds_barrier();
#pragma unroll
        for (int u = 0; u < BM * 16 / 256; ++u) {
            const int e = t + 256 * u;
            // row-major planes: 16 consecutive lanes take the 16 pieces of one row (64-byte runs per plane); slice-major planes:
            // 128 consecutive threads take the 128 rows of one piece
            const int row = g.out_sm ? (e & 127) : (e >> 4), c4 = (g.out_sm ? (e >> 7) : (e & 15)) * 4;
            const int64_t i = i0 + row;
            const int j = h2 * 64 + c4;
            if (i >= rows || j >= g.NB) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * CP + c4]);
            const int ct = g.Cb + g.Cbb;
            const int pl = j / ct, ch = j - pl * ct;
            if (ch < g.Cb) {
                if (pl == 0 && g.add0) {
                    const float4 e = *reinterpret_cast<const float4*>(g.add0 + i * g.Cb + ch);
                    v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
                }
                *reinterpret_cast<float4*>(plane_piece(g.out, pl, i, ch, g.Cb, g.M, g.out_sm, g.Cb)) = v;
            } else {
                *reinterpret_cast<float4*>(plane_piece(g.outb, pl, i, ch - g.Cb, g.Cbb, g.M, g.out_sm, g.Cbb)) = v;
            }
        }
    }
}

// ---- the whole backward pass of one gate-cell use in ONE persistent launch (hidden 8 / 16): cell backward, the data gradient
// gT = gG W^T AND the weight gradient gW = [T_0 .. T_{K-1} | S]^T gG.  The gate gradients gG (N, 4h) never exist in memory:
// a workgroup (one per CU, 512 threads) walks its share of the 128-row tiles, computes a tile's gG rows into LDS, feeds both
// MFMA products from there and keeps its partial gW (<= 128 x 64) in accumulator registers across all its tiles -- one slab
// per workgroup at the end, summed over the workgroups (and over the uses of the weight in the pass) by qt_colsum, in a
// fixed order.  Against qt_lstm_bwd_dgrad + the deferred qt_wgrad_group this drops the gG round trip (31 MB written and
// read back per use at the bench shape) and the separate weight-gradient launches, and the weight gradient's left operand
// is read while it is still warm from nothing -- it is read once either way -- but beside the cell's own traffic.
//   per tile:  TZ tile (128 x K) global -> LDS, row major (8 float4 in flight per thread)
//              cell backward of the 128 nodes (h / 4 lanes per node, k_lstm_bwd's arithmetic) -> gG tile in LDS, gCprev
//              barrier
//              wave w: weight-gradient tile (i block w & 3, j block w >> 2): 64 x mfma_32x32x2 over the 128 rows
//                      data-gradient tiles of row group w & 3 (column tiles split between waves 0-3 and 4-7)
//              barrier; data-gradient tiles -> LDS (over the TZ tile) -> row-contiguous float4 stores; barrier
struct CellBwdFusedArgs {
    DgradCellArgs d;          // cell operands, gCprev, part, BT = Wrows, NB, Kb, Cb, Cbb, out, outb, M (capacity), n_dev; gG unused
    PlaneSrc A;               // [T_0 .. T_{K-1} | S] of the forward pass
    int Kt;                   // rows of W: K * C + padded bias rows (<= 128)
    float* slab;              // (gridDim.x, Kt, 4h): this launch ADDS its partial weight gradients (zeroed by the caller)
};

template <int LPN, int NW, int NACC>
__device__ __forceinline__ void block_param_reduce_n(float (&acc)[NACC][4], int h, float* sm, float* part_row, int accumulate) {
    using namespace qtcell;
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
    __syncthreads();
    for (int idx = threadIdx.x; idx < NACC * h; idx += 64 * NW) {
        const int a = idx / h, j = idx % h;
        const int li = j >> 2, k = j & 3;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sm[(w * LPN + li) * NACC * 4 + a * 4 + k];
        part_row[idx] = accumulate ? part_row[idx] + s : s;
    }
}

// TR = rows per tile (threads = 4 TR): 64 -> two 256-thread workgroups per CU whose phases (loads + cell arithmetic / MFMA /
// stores) drift apart and overlap; 128 -> one 512-thread workgroup per CU (every phase of the CU in lockstep: 71 us per launch
// at the bench shape against 68 us for the separate launches it replaces).
template <int NT, int LPN, int TR>
__global__ __launch_bounds__(4 * TR, 2) void k_cell_bwd_fused(CellBwdFusedArgs f) {
    using namespace qtcell;
    const DgradCellArgs& g = f.d;
    constexpr int h = 4 * LPN, G4 = 4 * h, GP = G4 + 4, TP = 128, NJB = G4 / 32, NTA = (NT + 1) / 2;
    constexpr int NTHR = 4 * TR, NWAVE = NTHR / 64, NRG = TR / 32;            // waves = 2 NRG: (row group, column-tile half)
    constexpr int NWT = (4 * NJB + NWAVE - 1) / NWAVE;                        // weight-gradient tiles per wave
    constexpr int RSTEP = NTHR / 32, NU = TR / RSTEP;                         // TZ quads per thread (8)
    __shared__ __attribute__((aligned(16))) float TZt[TR * TP];       // TZ tile [row][k]; later the data-gradient staging tile
    __shared__ __attribute__((aligned(16))) float Gt[TR * GP];        // gG tile [row][4h]
    __shared__ __attribute__((aligned(16))) float Bt[32 * NT * GP];   // Wrows [column][4h]
    __shared__ const float* qptr[MAXQ];
    __shared__ int qstr[MAXQ];
    __shared__ float sm[NWAVE * LPN * 11 * 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    const int rows = qt_rows(g.n_dev, g.M);
    const int ntiles = (rows + TR - 1) / TR;
    const int t0 = (int)((int64_t)ntiles * blockIdx.x / gridDim.x), t1 = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    float* part_row = g.part + (int64_t)blockIdx.x * 11 * h;
    if (t0 >= t1) {                 // no tile for this workgroup: its slab rows keep what they hold (the launch only adds)
        if (!g.accumulate)
            for (int idx = t; idx < 11 * h; idx += NTHR) part_row[idx] = 0.0f;
        return;
    }
    const int nquad = f.Kt >> 2;
    for (int Q = t; Q < nquad; Q += NTHR) {              // (build_quad_table strides by 256 threads)
        const PlaneSrc& A = f.A;
        const int ct = A.Ca + A.Cab;
        const int k = 4 * Q, kc = A.Ka * ct;
        if (k < kc) {
            const int pl = k / ct, c = k - pl * ct;
            if (c < A.Ca) {
                if (pl > 0 && A.sm) {
                    qptr[Q] = A.a_rest + ((int64_t)(pl - 1) * (A.Ca / 4) + c / 4) * A.N * 4;
                    qstr[Q] = 4;
                } else {
                    qptr[Q] = (pl == 0 ? A.a0 : A.a_rest + (int64_t)(pl - 1) * A.N * A.Ca) + c;
                    qstr[Q] = pl == 0 ? A.lda0 : A.Ca;
                }
            } else {
                if (pl > 0 && A.sm) {
                    qptr[Q] = A.a_restb + ((int64_t)(pl - 1) * (A.Cab / 4) + (c - A.Ca) / 4) * A.N * 4;
                    qstr[Q] = 4;
                } else {
                    qptr[Q] = (pl == 0 ? A.a0b : A.a_restb + (int64_t)(pl - 1) * A.N * A.Cab) + (c - A.Ca);
                    qstr[Q] = pl == 0 ? A.lda0b : A.Cab;
                }
            }
        } else {
            qptr[Q] = A.S + (k - kc);
            qstr[Q] = A.Ks;
        }
    }
    for (int e = t; e < 32 * NT * (G4 / 4); e += NTHR) {
        const int c = e / (G4 / 4), kq = e - c * (G4 / 4);
        float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < g.NB) w = *reinterpret_cast<const float4*>(g.BT + (int64_t)c * G4 + 4 * kq);
        *reinterpret_cast<float4*>(&Bt[c * GP + 4 * kq]) = w;
    }
    const int j0 = (t % LPN) * 4;
    float pacc[11][4];
#pragma unroll
    for (int a = 0; a < 11; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) pacc[a][k] = 0.0f;
    f32x16 accw[NWT];               // this wave's tiles of the partial weight gradient
#pragma unroll
    for (int v = 0; v < NWT; ++v)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[v][r] = 0.0f;
    const int rg = wave % NRG, own = wave / NRG;
    __syncthreads();                // quad table, Bt

    const int Qq = t & 31, rb = t >> 5;
    const bool qok = Qq < nquad;
    const float* qp = qptr[qok ? Qq : 0];
    const int qs = qstr[qok ? Qq : 0];
    const int crow = t / LPN;
    float4 tz[NU];
    auto load_tz = [&](int tile) {
        const int64_t i0 = (int64_t)tile * TR;
        // thread (Q = t & 31, row = (t >> 5) + RSTEP u); quads beyond K and rows beyond the valid count are zeros (the capacity
        // rows of a static-mode operand hold garbage); the loads themselves are unconditional (clamped addresses)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int64_t row = i0 + rb + RSTEP * u;
            const bool ok = qok && row < rows;
            const float4 v = gload4(qp + (row < rows ? row : (int64_t)rows - 1) * qs);
            tz[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    };
    load_tz(t0);
    for (int tile = t0; tile < t1; ++tile) {
        const int64_t i0 = (int64_t)tile * TR;
        // (1) cell backward of the TR nodes (LPN lanes each) -> gG tile; the TZ quads (requested one tile ahead) -> LDS while
        // the cell operands are on their way
        {
            const int64_t node = i0 + crow;
            const bool act = t < TR * LPN, ok = act && node < rows;
            const F4 z = {{0, 0, 0, 0}};
            F4 I = z, F = z, T = z, Og = z, cp = z, gyh = z, gyc = z, go_in = z;
            if (ok) {
                const float* gs = g.gates + node * 4 * h + j0;
                I = ld4(gs); F = ld4(gs + h); T = ld4(gs + 2 * h); Og = ld4(gs + 3 * h);
                if (g.Cprev) cp = ld4(g.Cprev + node * g.ld_c + j0);
                if (g.gHn) gyh = ld4(g.gHn + node * g.ld_gh + j0);
                if (g.gCn) gyc = ld4(g.gCn + node * g.ld_gc + j0);
                if (g.gO) go_in = ld4(g.gO + node * g.ld_go + j0);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) *reinterpret_cast<float4*>(&TZt[(rb + RSTEP * u) * TP + 4 * Qq]) = tz[u];
            if (act) {
                const F4 wci = ld4(g.wc + j0), wcf = ld4(g.wc + h + j0), wco = ld4(g.wc + 2 * h + j0);
                F4 gam_h = {{1, 1, 1, 1}}, gam_c = {{1, 1, 1, 1}};
                if (g.ln) {
                    gam_h = ld4(g.ln + j0);
                    gam_c = ld4(g.ln + 2 * h + j0);
                }
                const CellBwdOut o = cell_backward<LPN>(I, F, T, Og, cp, gyh, gyc, go_in, wci, wcf, wco, gam_h, gam_c,
                                                        g.ln != nullptr, h, pacc);
                float* as = Gt + crow * GP + j0;
                st4(as, o.ggi); st4(as + h, o.ggf); st4(as + 2 * h, o.ggc); st4(as + 3 * h, o.ggo);
                if (ok && g.gCprev) st4(g.gCprev + node * h + j0, o.gcp);
            }
        }
        __syncthreads();            // TZt, Gt complete
        if (tile + 1 < t1) load_tz(tile + 1);          // in flight during the MFMA phase
        // (2) weight gradient: rows 2 s + half of the tile are the two k slots of step s; the operands of the next 4 steps
        // are read from LDS before the current 4 MFMAs issue
#pragma unroll
        for (int v = 0; v < NWT; ++v) {
            const int tau = wave + NWAVE * v, ib = tau & 3, jb = tau >> 2;
            if (jb < NJB) {
                const float* ap = TZt + half * TP + 32 * ib + l32;
                const float* bp = Gt + half * GP + 32 * jb + l32;
                float av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = ap[2 * q * TP]; bv[q] = bp[2 * q * GP]; }
#pragma unroll 1
                for (int s0 = 0; s0 < TR / 2; s0 += 4) {
                    float an[4], bn[4];
                    const int sn = s0 + 4 < TR / 2 ? s0 + 4 : s0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { an[q] = ap[2 * (sn + q) * TP]; bn[q] = bp[2 * (sn + q) * GP]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) accw[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], accw[v], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { av[q] = an[q]; bv[q] = bn[q]; }
                }
            }
        }
        // (3) data gradient of row group rg, one column tile per pass: the first half of the waves takes tile 2 ps, the second
        // half tile 2 ps + 1; the tile goes through LDS (over the TZ tile, once every wave is done with it) so that rows leave
        // as float4 pieces
        float* Cs = TZt;
#pragma unroll
        for (int ps = 0; ps < NTA; ++ps) {
            const int nt = 2 * ps + own;
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
            if (nt < NT) {
#pragma unroll
                for (int j = 0; j < G4 / 8; ++j) {
                    const float4 a = *reinterpret_cast<const float4*>(&Gt[(rg * 32 + l32) * GP + 8 * j + 4 * half]);
                    const float4 b = *reinterpret_cast<const float4*>(&Bt[(nt * 32 + l32) * GP + 8 * j + 4 * half]);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);
                }
            }
            __syncthreads();        // pass 0: every wave is done with the TZ tile; later passes: the stores have read the staging tile
            if (nt < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Cs[(rg * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 64 + own * 32 + l32] = acc2[r];
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = t + NTHR * u;
                const int row = e >> 4, c4 = (e & 15) * 4;
                const int64_t i = i0 + row;
                const int j = ps * 64 + c4;
                if (i < rows && j < g.NB) {
                    const float4 v = *reinterpret_cast<const float4*>(&Cs[row * 64 + c4]);
                    const int ct = g.Cb + g.Cbb;
                    const int pl = j / ct, ch = j - pl * ct;
                    if (ch < g.Cb)
                        *reinterpret_cast<float4*>(plane_piece(g.out, pl, i, ch, g.Cb, g.M, g.out_sm, g.Cb)) = v;
                    else
                        *reinterpret_cast<float4*>(plane_piece(g.outb, pl, i, ch - g.Cb, g.Cbb, g.M, g.out_sm, g.Cbb)) = v;
                }
            }
        }
        __syncthreads();            // the stores have read the staging tile / Gt is free: the next tile may overwrite both
    }
    // partial weight gradient of this workgroup: added to its slab
#pragma unroll
    for (int v = 0; v < NWT; ++v) {
        const int tau = wave + NWAVE * v, ib = tau & 3, jb = tau >> 2;
        if (jb < NJB) {
            float* sl = f.slab + (int64_t)blockIdx.x * f.Kt * G4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * ib + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < f.Kt) sl[(int64_t)row * G4 + 32 * jb + l32] += accw[v][r];
            }
        }
    }
    block_param_reduce_n<LPN, NWAVE, 11>(pacc, h, sm, part_row, g.accumulate);
}

}  // namespace

// shared argument checks / operand setup of the node-feature operand
static int plane_src(PlaneSrc* A, const char* fn, const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b,
                     const float* a_restb, int Ka, int Ca, int Cab, const float* S, int Ks, int N, int sm = 0) {
    const bool ok = a0 && Ka >= 1 && Ca >= 1 && Cab >= 0 && (Ka == 1 || a_rest) && (Cab == 0 || (a0b && (Ka == 1 || a_restb))) &&
                    (Ks == 0 || S) && Ca % 4 == 0 && Cab % 4 == 0 && Ks % 4 == 0 && lda0 % 4 == 0 && lda0b % 4 == 0 && (Ka * (Ca + Cab) + Ks) / 4 <= MAXQ &&
                    (((uintptr_t)a0 | (uintptr_t)a_rest | (uintptr_t)a0b | (uintptr_t)a_restb | (uintptr_t)S) & 15) == 0;
    if (!ok) {
        qt_set_error("%s: bad node-feature operand (planes / parts must be 16-byte aligned with widths that are multiples of 4, "
                     "reduction dimension <= 512)", fn);
        return QT_E_ARG;
    }
    A->a0 = a0; A->a_rest = a_rest; A->a0b = Cab ? a0b : nullptr; A->a_restb = Cab ? a_restb : nullptr; A->S = S;
    A->Ka = Ka; A->Ca = Ca; A->Cab = Cab; A->Ks = Ks; A->N = N;
    A->lda0 = lda0 > 0 ? lda0 : Ca; A->lda0b = lda0b > 0 ? lda0b : Cab;
    A->sm = sm != 0;
    return QT_OK;
}

#ifdef QT_GEMM_TIMING
static long long* g_dbg = nullptr;       // diagnostics build only (tools/exp_gemm_timing.py)
extern "C" void qt_gemm_timing_buffer(long long* p) { g_dbg = p; }
#endif

static int spmm_part(SpmmPart* P, int* nblk, int N, int C, const float* x, int ldx, const float* p, int ldp, const float* q,
                     int ldq, float* out) {
    P->x = x; P->p = p; P->q = q; P->out = out; P->C = C;
    P->ldx = ldx > 0 ? ldx : C; P->ldp = ldp > 0 ? ldp : C; P->ldq = ldq > 0 ? ldq : C;
    int grid = qt_cdiv((int64_t)N * (C / 4), QT_SPMM_BS);
    static const bool xcd = getenv("QT_SPMM_FLAT") == nullptr;
    P->xcd_chunk = 0;
    if (xcd && grid >= 64) {
        P->xcd_chunk = qt_cdiv(grid, 8);
        grid = P->xcd_chunk * 8;    // surplus workgroups fall past the row count and exit
    }
    *nblk = grid;
    return 0;
}

extern "C" int qt_spmm2(const int32_t* rowptr, const int32_t* col, const float* nrm, int N, const int32_t* n_dev, int Ca,
                        const float* xa, int ldxa, const float* pa, int ldpa, const float* qa, int ldqa, float* outa, int Cb,
                        const float* xb, int ldxb, const float* pb, int ldpb, const float* qb, int ldqb, float* outb,
                        float alpha, float beta, float gamma, const int32_t* ell, void* stream) {
    QT_ARG((ldxa | ldpa | ldqa | ldxb | ldpb | ldqb) % 4 == 0, "row strides must be multiples of 4");
    QT_ARG(rowptr && col && nrm && xa && outa && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0, "bad arguments");
    QT_ARG(Cb == 0 || (xb && outb && (pb != nullptr) == (pa != nullptr) && (qb != nullptr) == (qa != nullptr)),
           "part b must mirror part a");
    QT_ARG(xa != outa && (Cb == 0 || xb != outb), "out must not alias x");
    QT_ARG((int64_t)N * max(Ca, Cb) / 4 + 2048 < (int64_t)1 << 31, "N * C too large for 32-bit thread indices");
    QT_ARG((((uintptr_t)xa | (uintptr_t)outa | (uintptr_t)pa | (uintptr_t)qa | (uintptr_t)xb | (uintptr_t)outb | (uintptr_t)pb |
             (uintptr_t)qb | (uintptr_t)ell) & 15) == 0, "operands must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    SpmmPart A, B = {};
    int na = 0, nb = 0;
    spmm_part(&A, &na, N, Ca, xa, ldxa, pa, ldpa, qa, ldqa, outa);
    if (Cb) spmm_part(&B, &nb, N, Cb, xb, ldxb, pb, ldpb, qb, ldqb, outb);
    // 8 edges per trip for narrow rows (see qt_spmm); the wider part decides
#ifndef QT_EPT8_MAXC
#define QT_EPT8_MAXC 20
#endif
    if (max(Ca, Cb) <= QT_EPT8_MAXC)
        hipLaunchKernelGGL((k_spmm<4, 1, 8>), dim3(na + nb), dim3(QT_SPMM_BS), 0, (hipStream_t)stream, rowptr, col, nrm, reinterpret_cast<const int4*>(ell), N, n_dev, A, B, na, alpha, beta, gamma);
    else
        hipLaunchKernelGGL((k_spmm<4, 1, 4>), dim3(na + nb), dim3(QT_SPMM_BS), 0, (hipStream_t)stream, rowptr, col, nrm, reinterpret_cast<const int4*>(ell), N, n_dev, A, B, na, alpha, beta, gamma);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_spmm1(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell, int N, const int32_t* n_dev,
                        const float* x, int ldx, float alpha, const float* p, int ldp, float beta, const float* q, int ldq,
                        float gamma, float* out, int ldo, int pad4, int act, const float* res, int ldr, const float* drop,
                        void* stream) {
    QT_ARG(rowptr && col && nrm && x && out && ldx >= 1 && ldo >= 1 && (!p || ldp >= 1) && (!q || ldq >= 1), "bad arguments");
    QT_ARG(act == QT_ACT_NONE || (act == QT_ACT_TANH_RES && res && ldr >= 1), "the epilogue is none or tanh(drop v) + res");
    QT_ARG(!pad4 || (ldo % 4 == 0 && ((uintptr_t)out & 15) == 0), "pad4 writes 16-byte rows");
    QT_ARG(((uintptr_t)ell & 15) == 0, "ell must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    Spmm1Args a = {rowptr, col, nrm, reinterpret_cast<const int4*>(ell), N, n_dev, x, p, q, ldx, ldp, ldq, alpha, beta, gamma,
                   out, ldo, pad4, act, res, ldr, drop};
    const int grid = qt_cdiv(qt_cdiv(N, 64), 8) * 8;
    hipLaunchKernelGGL(k_spmm1, dim3(grid), dim3(64), 0, (hipStream_t)stream, a);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_spmm(const int32_t* rowptr, const int32_t* col, const float* nrm, int N, const int32_t* n_dev, int C,
                       const float* x, float alpha, const float* p, float beta, const float* q, float gamma, float* out,
                       void* stream) {
    QT_ARG(rowptr && col && nrm && x && out && C > 0, "bad arguments");
    QT_ARG(x != out, "out must not alias x");
    if (N <= 0) return QT_OK;
    const bool v4 = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)p | (uintptr_t)q) % 16) == 0);
    if (v4) return qt_spmm2(rowptr, col, nrm, N, n_dev, C, x, 0, p, 0, q, 0, out, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, alpha, beta, gamma, nullptr, stream);
    // scalar rows (C not a multiple of 4): one float per thread
    SpmmPart A, B = {};
    A.x = x; A.p = p; A.q = q; A.out = out; A.C = C; A.ldx = A.ldp = A.ldq = C; A.xcd_chunk = 0;
    const int grid = qt_cdiv((int64_t)N * C, QT_SPMM_BS);
    // Edges per trip: a trip is two dependent loads (col/nrm, then the x rows), and the few rows with many neighbours
    // (a 4x4 cell next to 1x1 cells has 16) set the length of the whole launch.  8 per trip: 6.6 -> 4.5 us at C = 4,
    // 9.3 -> 7.5 us at C = 16 (N = 1.2e5, inside a hipGraph); wider rows are bandwidth bound and prefer fewer registers.
    hipLaunchKernelGGL((k_spmm<1, 1, 8>), dim3(grid), dim3(QT_SPMM_BS), 0, (hipStream_t)stream, rowptr, col, nrm, (const int4*)nullptr, N, n_dev, A, B, grid, alpha, beta, gamma);
    QT_LAUNCHED();
    return QT_OK;
}

// Tile width of a plain product by its output width: the fewest 32-column MFMA tiles over all column blocks (NB = 280: 3 blocks of
// 3 tiles = 9 tiles instead of 3 x 4 = 12), ties to the wider block (A is re-read per block).
static inline int gemm_nt(int NB) {
    int best = 4, cost = qt_cdiv(NB, 128) * 4;
    for (int nt = 3; nt >= 2; --nt) {
        const int c = qt_cdiv(NB, 32 * nt) * nt;
        if (c < cost) {
            best = nt;
            cost = c;
        }
    }
    return NB <= 64 ? 2 : best;
}
#ifndef QT_GEMM_KWT3
#define QT_GEMM_KWT3 128
#endif
#ifndef QT_GEMM_KWT4
#define QT_GEMM_KWT4 64      // k rows of W staged per pass by the 128-column tiles (NT = 4)
#endif
static void launch_gemm_fwd(const GemmArgs& g, int N, int G, hipStream_t stream) {
    switch (gemm_nt(g.NB)) {
        case 2: hipLaunchKernelGGL((k_gemm_fwd<2, 128>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 64), G), dim3(256), 0, stream, g); break;
        case 3: hipLaunchKernelGGL((k_gemm_fwd<3, QT_GEMM_KWT3>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 96), G), dim3(256), 0, stream, g); break;
        default: hipLaunchKernelGGL((k_gemm_fwd<4, QT_GEMM_KWT4>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 128), G), dim3(256), 0, stream, g); break;
    }
}

extern "C" int qt_dense2(const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb, int Ka,
                         int Ca, int Cab,
                         const float* W, const float* WT, const float* S, int Ks, const float* Ws, int Kb, int Cb, int Cbb, int N,
                         const int32_t* n_dev, int act, const float* res, int res_stride, const float* drop, float* out,
                         float* outb, int planes_sm, const float* post_W, float* post_out, void* stream) {
    QT_ARG((W || WT) && out && Kb >= 1 && Cb >= 1 && Cbb >= 0 && (Cbb == 0 || outb), "bad arguments");
    QT_ARG((post_W == nullptr) == (post_out == nullptr) && (!post_W || (W && Kb * (Cb + Cbb) == 16 && (((uintptr_t)post_W | (uintptr_t)post_out) & 15) == 0)),
           "the second product needs 16 output columns, W (not WT) and 16-byte aligned post_W (20, 4) / post_out (N, 4)");
    QT_ARG(act != QT_ACT_RELU_BWD || (res && res_stride >= Kb * (Cb + Cbb) && res_stride % 4 == 0 && W && Kb * (Cb + Cbb) <= 16 && Kb * (Cb + Cbb) > 4),
           "QT_ACT_RELU_BWD: res = the forward output (N, res_stride), 8 .. 16 output columns, W (not WT)");
    QT_ARG((Ks == 0) || Ws || WT, "Ws missing");
    QT_ARG(Ks == 0 || WT || Ws == W + (int64_t)Ka * (Ca + Cab) * Kb * (Cb + Cbb), "Ws must follow W contiguously ([W ; Ws] is one matrix)");
    QT_ARG(act == QT_ACT_NONE || (Kb == 1 && Cbb == 0), "activation needs one undivided output plane");
    QT_ARG(Cb % 4 == 0 && Cbb % 4 == 0, "Cb and Cbb must be multiples of 4 (float4 stores)");
    QT_ARG((((uintptr_t)W | (uintptr_t)WT) & 15) == 0, "W / WT must be 16-byte aligned");
    QT_ARG(act != QT_ACT_TANH_RES || res, "QT_ACT_TANH_RES needs res");
    GemmArgs g = {};
    if (int rc = plane_src(&g.A, __func__, a0, lda0, a_rest, a0b, lda0b, a_restb, Ka, Ca, Cab, S, Ks, N, planes_sm & 1)) return rc;
    g.out_sm = (planes_sm >> 1) & 1;
    if (N <= 0) return QT_OK;
    g.B = W; g.BT = WT; g.M = N; g.K = Ka * (Ca + Cab) + Ks; g.NB = Kb * (Cb + Cbb);
    g.outb = outb; g.Cbb = Cbb;
#ifdef QT_GEMM_TIMING
    g.dbg = g_dbg;
#endif
    g.Kb = Kb; g.Cb = Cb; g.act = act; g.res = res; g.res_stride = res_stride; g.drop = drop; g.out = out; g.row0_step = 0; g.n_dev = n_dev; g.accumulate = 0;
    g.post_W = post_W; g.post_out = post_out;
    // default: exact fp32 MFMA (bit-for-bit a k-ordered fmaf chain).  QT_GEMM_BF16X3=1 opts into the bf16x3 split
    // kernels (fp32-level error, ~8 % faster on these memory/latency-shaped GEMMs: measured 27.7 vs 30.1 us).
    static const bool exact_fp32 = getenv("QT_GEMM_BF16X3") == nullptr;
    static const bool no_skinny = getenv("QT_GEMM_NO_SKINNY") != nullptr;
    if ((!no_skinny || post_W || act == QT_ACT_RELU_BWD) && g.NB <= 16 && W) {       // (wide outputs of short reductions measured slower here: 24.5 vs 14.5 us)
        static const bool no_row16 = getenv("QT_GEMM_NO_ROW16") != nullptr;       // (A/B switch)
        if (g.NB <= 4)
            hipLaunchKernelGGL((k_gemm_skinny<256>), dim3(qt_cdiv(N, 256), 1, 1), dim3(256), 0, (hipStream_t)stream, g);
        else if (!no_row16 && g.NB == 16 && Kb == 1 && Cbb == 0 && act != QT_ACT_TANH_RES && g.K <= 256)
            hipLaunchKernelGGL(k_gemm_row16, dim3(qt_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL((k_gemm_skinny<64>), dim3(qt_cdiv(N, 64), qt_cdiv(g.NB, 16), 1), dim3(256), 0, (hipStream_t)stream, g);
    } else if (exact_fp32) {
        launch_gemm_fwd(g, N, 1, (hipStream_t)stream);
    } else {
        if (g.NB > 64)
            hipLaunchKernelGGL((k_gemm_fwd3<4, 64>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 128), 1), dim3(256), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL((k_gemm_fwd3<2, 128>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 64), 1), dim3(256), 0, (hipStream_t)stream, g);
    }
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_head_dgrad(const float* gU, const float* Wb2, const float* Y, const float* Wb1, int K, int Cb, int Cbb, int N,
                             const int32_t* n_dev, float* G, float* out, float* outb, int out_sm, void* stream) {
    QT_ARG(gU && Wb2 && Y && Wb1 && G && out && K >= 1 && Cb > 0 && Cb % 4 == 0 && Cbb >= 0 && Cbb % 4 == 0 && (Cbb == 0 || outb), "bad arguments");
    QT_ARG((((uintptr_t)gU | (uintptr_t)Wb2 | (uintptr_t)Y | (uintptr_t)Wb1 | (uintptr_t)G | (uintptr_t)out | (uintptr_t)outb) & 15) == 0,
           "operands must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    HeadDgradArgs g = {gU, Wb2, Y, Wb1, G, out, outb, N, K, Cb, Cbb, out_sm != 0, n_dev};
    hipLaunchKernelGGL(k_head_dgrad, dim3(qt_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, g);
    QT_LAUNCHED();
    return QT_OK;
}

// G independent products in one launch (grid z = group): group g multiplies [A_g | S], A_g = Ka planes (N, Ca) starting at
// A + g gsA, with W_g and writes Kb planes (N, Cb) starting at out + g gsO (row stride ldo).
// The eight GraphConv stacks of a GConvLSTM with attention convolutions (model/model.py:394-424) run layer by layer: group g
// is stack g's projection [q | k | v | skip], read from and written to blocks of arrays shared by all stacks.
extern "C" int qt_proj_group(const float* A, int lda, int64_t gsA, int Ka, int Ca, const float* S, const float* W, const float* WT,
                             int64_t gsW, int G, int Kb, int Cb, float* out, int ldo, int64_t gsO, int reverse, int N,
                             const int32_t* n_dev, void* stream) {
    QT_ARG((W || WT) && out && G >= 1 && G <= 65535 && Kb >= 1 && Cb >= 4 && Cb % 4 == 0 && Ka >= 1, "bad arguments");
    if (ldo == 0) ldo = Cb;
    QT_ARG(ldo >= Cb && ldo % 4 == 0 && gsA % 4 == 0 && gsW % 4 == 0 && gsO % 4 == 0, "strides must be multiples of 4 floats");
    QT_ARG(Ka == 1 || lda == 0 || lda == Ca, "several input planes must be dense");
    QT_ARG((((uintptr_t)W | (uintptr_t)WT | (uintptr_t)out) & 15) == 0, "W / WT / out must be 16-byte aligned");
    GemmArgs g = {};
    if (int rc = plane_src(&g.A, __func__, A, lda, Ka > 1 ? A + (int64_t)N * Ca : nullptr, nullptr, 0, nullptr, Ka, Ca, 0, S, S ? 4 : 0, N))
        return rc;
    if (N <= 0) return QT_OK;
    g.B = W; g.BT = WT; g.M = N; g.K = Ka * Ca + (S ? 4 : 0); g.NB = Kb * Cb;
    g.Kb = Kb; g.Cb = Cb; g.act = QT_ACT_NONE; g.out = out; g.n_dev = n_dev;
    g.ldo = ldo; g.gsA = gsA; g.gsB = gsW; g.gsO = gsO;
    g.zrev = reverse != 0;
#ifdef QT_GEMM_TIMING
    g.dbg = nullptr;
#endif
    launch_gemm_fwd(g, N, G, (hipStream_t)stream);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_dense_sb(const float* A, int lda, int K, const void* Whi, const void* Wlo, int Kb, int Cb, int Cbb, int N,
                           const int32_t* n_dev, float* out, float* outb, void* stream) {
    QT_ARG(A && Whi && Wlo && out && Kb >= 1 && Cb >= 4 && Cb % 4 == 0 && Cbb >= 0 && Cbb % 4 == 0 && (Cbb == 0 || outb), "bad arguments");
    if (lda == 0) lda = K;
    QT_ARG(K >= 16 && K % 16 == 0 && lda >= K && lda % 4 == 0, "the reduction length must be a multiple of 16");
    QT_ARG((((uintptr_t)A | (uintptr_t)Whi | (uintptr_t)Wlo | (uintptr_t)out | (uintptr_t)outb) & 15) == 0, "operands must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    GemmArgs g = {};
    g.A.a0 = A; g.A.lda0 = lda; g.A.N = N; g.A.Ka = 1; g.A.Ca = K;
    g.M = N; g.K = K; g.NB = Kb * (Cb + Cbb); g.Kb = Kb; g.Cb = Cb; g.Cbb = Cbb; g.out = out; g.outb = outb; g.n_dev = n_dev;
    if (g.NB > 64)
        hipLaunchKernelGGL((k_gemm_sb<4>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 128)), dim3(256), 0, (hipStream_t)stream, g, (const __bf16*)Whi,
                           (const __bf16*)Wlo);
    else
        hipLaunchKernelGGL((k_gemm_sb<2>), dim3(qt_cdiv(N, BM), qt_cdiv(g.NB, 64)), dim3(256), 0, (hipStream_t)stream, g, (const __bf16*)Whi,
                           (const __bf16*)Wlo);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_dense(const float* a0, const float* a_rest, int Ka, int Ca, const float* W, const float* S, int Ks,
                        const float* Ws, int Kb, int Cb, int N, const int32_t* n_dev, int act, const float* res,
                        int res_stride, const float* drop, float* out, void* stream) {
    return qt_dense2(a0, 0, a_rest, nullptr, 0, nullptr, Ka, Ca, 0, W, nullptr, S, Ks, Ws, Kb, Cb, 0, N, n_dev, act, res, res_stride, drop, out,
                     nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int qt_lstm_dgrad_blocks(int N) { return N <= 0 ? 0 : qt_cdiv(N, BM); }

extern "C" int qt_lstm_bwd_dgrad(const float* gO, int ld_go, const float* gHn, int ld_gh, const float* gCn, int ld_gc,
                                 const float* gates, const float* Cprev, int ld_c, const float* wc, const float* ln, int N,
                                 const int32_t* n_dev, int h, float* gG, float* gCprev, float* part, int accumulate,
                                 const float* Wrows, const void* Whi, const void* Wlo, int Kb, int Cb, int Cbb, float* out,
                                 float* outb, int out_sm, const float* gHn2, int ld_gh2, const float* add0, void* stream) {
    QT_ARG(gates && wc && gG && part && Wrows && out, "null pointer");
    QT_ARG((!gHn2 || (ld_gh2 >= h && ld_gh2 % 4 == 0)) && (((uintptr_t)gHn2 | (uintptr_t)add0) & 15) == 0, "bad second gradient / plane-0 addend");
    QT_ARG((Whi == nullptr) == (Wlo == nullptr) && (((uintptr_t)Whi | (uintptr_t)Wlo) & 15) == 0, "Whi / Wlo come as a 16-byte aligned pair");
    QT_ARG(h == 8 || h == 16, "fused for hidden sizes 8 and 16 (others: qt_lstm_bwd + qt_dense2)");
    QT_ARG(Kb >= 1 && Cb >= 4 && Cb % 4 == 0 && Cbb >= 0 && Cbb % 4 == 0 && (Cbb == 0 || outb), "bad output planes");
    const int NB = Kb * (Cb + Cbb);
    QT_ARG(NB <= 128, "the output planes must fit one 128-column tile");
    QT_ARG((!gHn || ld_gh >= h) && (!gCn || ld_gc >= h) && (!gO || ld_go >= h) && ld_gh % 4 == 0 && ld_gc % 4 == 0 &&
               ld_go % 4 == 0 && ld_c % 4 == 0 && (!Cprev || ld_c >= h), "bad row stride");
    QT_ARG((((uintptr_t)Wrows | (uintptr_t)gates | (uintptr_t)gG | (uintptr_t)out | (uintptr_t)outb) & 15) == 0, "operands must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    DgradCellArgs g = {};
    g.gO = gO; g.gHn = gHn; g.gCn = gCn; g.gates = gates; g.Cprev = Cprev; g.wc = wc; g.ln = ln;
    g.ld_go = ld_go; g.ld_gh = ld_gh; g.ld_gc = ld_gc; g.ld_c = ld_c; g.h = h;
    g.gG = gG; g.gCprev = gCprev; g.part = part; g.accumulate = accumulate;
    g.BT = Wrows; g.M = N; g.NB = NB; g.Kb = Kb; g.Cb = Cb; g.Cbb = Cbb; g.out = out; g.outb = outb; g.n_dev = n_dev;
    g.BThi = (const __bf16*)Whi; g.BTlo = (const __bf16*)Wlo;
    g.out_sm = out_sm != 0;
    g.gHn2 = gHn2; g.ld_gh2 = ld_gh2; g.add0 = add0;
    const dim3 grid(qt_cdiv(N, BM));
    // 32-column MFMA tiles: as many as the output planes need (K' C = 80 or 96 columns take three, not four)
    if (h == 16) {
        if (NB <= 64) hipLaunchKernelGGL((k_dgrad_cell<2, 4>), grid, dim3(256), 0, (hipStream_t)stream, g);
        else if (NB <= 96) hipLaunchKernelGGL((k_dgrad_cell<3, 4>), grid, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((k_dgrad_cell<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, g);
    } else {
        if (NB <= 64) hipLaunchKernelGGL((k_dgrad_cell<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
        else if (NB <= 96) hipLaunchKernelGGL((k_dgrad_cell<3, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
        else hipLaunchKernelGGL((k_dgrad_cell<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
    }
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_num_cus(void) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;         // (no device visible: the MI355X count; only sizes host-side buffers)
    }
    return n_cu;
}

extern "C" int qt_lstm_bwd_fused(const float* gO, int ld_go, const float* gHn, int ld_gh, const float* gCn, int ld_gc,
                                 const float* gates, const float* Cprev, int ld_c, const float* wc, const float* ln, int N,
                                 const int32_t* n_dev, int h, float* gCprev, float* part, int accumulate,
                                 const float* Wrows, int Kb, int Cb, int Cbb, float* out, float* outb,
                                 const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb,
                                 int Ka, int Ca, int Cab, const float* S, int Ks, float* slab, int nslab, void* stream) {
    QT_ARG(gates && wc && part && Wrows && out && slab, "null pointer");
    QT_ARG(h == 8 || h == 16, "fused for hidden sizes 8 and 16");
    QT_ARG(Kb >= 1 && Cb >= 4 && Cb % 4 == 0 && Cbb >= 0 && Cbb % 4 == 0 && (Cbb == 0 || outb), "bad output planes");
    const int NB = Kb * (Cb + Cbb);
    QT_ARG(NB <= (h == 16 ? 128 : 64), "the output planes must fit the column tiles of the launch");
    QT_ARG((!gHn || ld_gh >= h) && (!gCn || ld_gc >= h) && (!gO || ld_go >= h) && ld_gh % 4 == 0 && ld_gc % 4 == 0 &&
               ld_go % 4 == 0 && ld_c % 4 == 0 && (!Cprev || ld_c >= h), "bad row stride");
    QT_ARG((((uintptr_t)Wrows | (uintptr_t)gates | (uintptr_t)out | (uintptr_t)outb | (uintptr_t)slab) & 15) == 0, "operands must be 16-byte aligned");
    CellBwdFusedArgs f = {};
    if (int rc = plane_src(&f.A, __func__, a0, lda0, a_rest, a0b, lda0b, a_restb, Ka, Ca, Cab, S, Ks, N)) return rc;
    f.Kt = Ka * (Ca + Cab) + Ks;
    QT_ARG(f.Kt <= 128, "the weight must have at most 128 rows (one accumulator tile column per wave)");
    const int NTc = qt_cdiv(NB, 32);
    // 64-row tiles, two 256-thread workgroups per CU (their LDS fits twice up to three column tiles); else 128-row tiles
    static const bool tr128 = getenv("QT_FUSED_TR128") != nullptr;
    const bool small = !tr128 && NTc <= 3;
    const int grid = small ? min(2 * qt_num_cus(), qt_cdiv(N, 64)) : min(qt_num_cus(), qt_cdiv(N, 128));
    QT_ARG(nslab >= grid, "slab too small: one (Kt, 4h) slab per workgroup, qt_lstm_fused_blocks() of them");
    if (N <= 0) return QT_OK;
    DgradCellArgs& g = f.d;
    g.gO = gO; g.gHn = gHn; g.gCn = gCn; g.gates = gates; g.Cprev = Cprev; g.wc = wc; g.ln = ln;
    g.ld_go = ld_go; g.ld_gh = ld_gh; g.ld_gc = ld_gc; g.ld_c = ld_c; g.h = h;
    g.gG = nullptr; g.gCprev = gCprev; g.part = part; g.accumulate = accumulate;
    g.BT = Wrows; g.M = N; g.NB = NB; g.Kb = Kb; g.Cb = Cb; g.Cbb = Cbb; g.out = out; g.outb = outb; g.n_dev = n_dev;
    f.slab = slab;
#define QT_FUSED(NT_, LPN_)                                                                                                  \
    do {                                                                                                                     \
        if (small) hipLaunchKernelGGL((k_cell_bwd_fused<NT_, LPN_, 64>), dim3(grid), dim3(256), 0, (hipStream_t)stream, f);  \
        else hipLaunchKernelGGL((k_cell_bwd_fused<NT_, LPN_, 128>), dim3(grid), dim3(512), 0, (hipStream_t)stream, f);       \
    } while (0)
    if (h == 16) {
        if (NTc <= 1) QT_FUSED(1, 4);
        else if (NTc == 2) QT_FUSED(2, 4);
        else if (NTc == 3) QT_FUSED(3, 4);
        else hipLaunchKernelGGL((k_cell_bwd_fused<4, 4, 128>), dim3(grid), dim3(512), 0, (hipStream_t)stream, f);
    } else {
        if (NTc <= 1) QT_FUSED(1, 2);
        else QT_FUSED(2, 2);
    }
#undef QT_FUSED
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_lstm_fused_blocks(void) { return 2 * qt_num_cus(); }

extern "C" int qt_dense_lstm(const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb,
                             int Ka, int Ca, int Cab, const float* W, const float* WT, const float* S, int Ks,
                             const float* Ws, int h, int N, const int32_t* n_dev, const float* Cprev, int ld_c,
                             const float* wc, const float* b, const float* ln, float* O, float* Hn, float* Cn,
                             float* gates, int planes_sm, void* stream) {
    QT_ARG((W || WT) && wc && b && Hn && Cn && gates, "bad arguments");
    QT_ARG(h == 8 || h == 16 || h == 32, "the fused gate GEMM + cell covers hidden sizes 8, 16 and 32 (qt_dense + qt_lstm_fwd otherwise)");
    QT_ARG((Ks == 0) || Ws || WT, "Ws missing");
    QT_ARG(Ks == 0 || WT || Ws == W + (int64_t)Ka * (Ca + Cab) * 4 * h, "Ws must follow W contiguously ([W ; Ws] is one matrix)");
    QT_ARG((((uintptr_t)W | (uintptr_t)WT | (uintptr_t)Cprev) & 15) == 0 && ld_c % 4 == 0, "operands must be 16-byte aligned");
    GemmArgs g = {};
    if (int rc = plane_src(&g.A, __func__, a0, lda0, a_rest, a0b, lda0b, a_restb, Ka, Ca, Cab, S, Ks, N, planes_sm)) return rc;
    if (N <= 0) return QT_OK;
    g.B = W; g.BT = WT; g.M = N; g.K = Ka * (Ca + Cab) + Ks; g.NB = 4 * h;
    g.Kb = 1; g.Cb = 4 * h; g.act = QT_ACT_NONE; g.res = nullptr; g.res_stride = 0; g.drop = nullptr; g.out = nullptr;
    g.row0_step = 0; g.n_dev = n_dev; g.accumulate = 0;
    g.Cprev = Cprev; g.wc = wc; g.bias = b; g.ln = ln; g.ld_c = ld_c; g.h = h;
    g.O = O; g.Hn = Hn; g.Cn = Cn; g.gates = gates;
#ifdef QT_GEMM_TIMING
    g.dbg = g_dbg;
#endif
    // hidden 8 / 16 with the whole W^T in LDS: the persistent wave-centric kernel (one workgroup per CU)
    static const bool persistent = getenv("QT_GATE_CELL_TILED") == nullptr;
    if (persistent && (h == 8 || h == 16) && g.K <= GATE_P_MAXK) {
        const int pitch = g.K + (((g.K >> 2) & 1) ? 8 : 4);                 // pitch / 4 odd
        const int n_cu = qt_num_cus();
        const dim3 pgrid(min(n_cu, qt_cdiv(N, 32)), 1, 1);
        const int nj = (g.K + 7) >> 3;
#define QT_GATE_P(NT_, LPN_)                                                                                              \
        do {                                                                                                              \
            if (nj == 8) hipLaunchKernelGGL((k_gate_cell_p<NT_, LPN_, 8>), pgrid, dim3(512), 0, (hipStream_t)stream, g, pitch);       \
            else if (nj == 11) hipLaunchKernelGGL((k_gate_cell_p<NT_, LPN_, 11>), pgrid, dim3(512), 0, (hipStream_t)stream, g, pitch); \
            else if (nj == 13) hipLaunchKernelGGL((k_gate_cell_p<NT_, LPN_, 13>), pgrid, dim3(512), 0, (hipStream_t)stream, g, pitch); \
            else hipLaunchKernelGGL((k_gate_cell_p<NT_, LPN_, 0>), pgrid, dim3(512), 0, (hipStream_t)stream, g, pitch);               \
        } while (0)
        if (h == 16) QT_GATE_P(2, 4);
        else QT_GATE_P(1, 2);
#undef QT_GATE_P
        QT_LAUNCHED();
        return QT_OK;
    }
    const dim3 grid(qt_cdiv(N, BM), 1, 1);
    if (h == 32)
        hipLaunchKernelGGL((k_gemm_fwd<4, QT_GEMM_KWT4, 8>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (h == 16)
        hipLaunchKernelGGL((k_gemm_fwd<2, 128, 4>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL((k_gemm_fwd<2, 128, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_wgrad_blocks(int N) { return N > 0 ? qt_cdiv(N, WGRAD_ROWS) : 0; }

extern "C" int qt_wgrad(const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb, int Ka,
                        int Ca, int Cab,
                        const float* S, int Ks, const float* G, int Co, int N, const int32_t* n_dev, int accumulate,
                        float* part, int planes_sm, void* stream) {
    QT_ARG(G && part && Co >= 1 && Co % 4 == 0 && ((uintptr_t)G & 15) == 0, "bad arguments");
    GemmArgs g = {};
    if (int rc = plane_src(&g.A, __func__, a0, lda0, a_rest, a0b, lda0b, a_restb, Ka, Ca, Cab, S, Ks, N, planes_sm)) return rc;
    if (N <= 0) return QT_OK;
    g.B = G; g.M = Ka * (Ca + Cab) + Ks; g.K = N; g.NB = Co;
    g.Kb = 1; g.Cb = Co; g.act = QT_ACT_NONE; g.res = nullptr; g.res_stride = 0; g.drop = nullptr; g.out = part;
    g.row0_step = WGRAD_ROWS;
    g.n_dev = n_dev;
    g.accumulate = accumulate;
    const dim3 grid(qt_cdiv(g.M, wgrad_fw(g.M) * 32), qt_cdiv(Co, BN), qt_cdiv(N, WGRAD_ROWS));
    QT_WGRAD_DISPATCH(k_gemm_wgrad, g.M, Co, grid, stream, g);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_wgrad_group_blocks(int nseg, const int* N) {
    int z = 0;
    for (int i = 0; i < nseg; ++i) z += N[i] > 0 ? qt_cdiv(N[i], WGRAD_GROUP_ROWS) : 0;
    return z;
}

static int wgrad_group_launch(const char* fn, int nseg, const float* const* a0, const int* lda0, const float* const* a_rest,
                              const float* const* a0b, const int* lda0b, const float* const* a_restb, const float* const* S,
                              const float* const* G, const int* N, const int32_t* const* n_dev, int Ka, int Ca, int Cab, int Ks,
                              int Co, int ldg, int gpl, int Gn, int64_t gsA, int64_t gsG, int per_node, float* part, void* stream, int sm = 0) {
    WgradGroup w;
    w.sm = sm != 0;
    int z = 0, k = 0;
    for (int i = 0; i < nseg; ++i) {
        if (N[i] <= 0) continue;
        PlaneSrc A;
        if (int rc = plane_src(&A, fn, a0[i], lda0 ? lda0[i] : 0, Ka > 1 ? a_rest[i] : nullptr, Cab ? a0b[i] : nullptr,
                               (Cab && lda0b) ? lda0b[i] : 0, (Cab && Ka > 1) ? a_restb[i] : nullptr, Ka, Ca, Cab,
                               Ks ? S[i] : nullptr, Ks, N[i]))
            return rc;
        if (!G[i] || ((uintptr_t)G[i] & 15) != 0) {
            qt_set_error("%s: G must be 16-byte aligned", fn);
            return QT_E_ARG;
        }
        w.a0[k] = A.a0; w.a_rest[k] = A.a_rest; w.a0b[k] = A.a0b; w.a_restb[k] = A.a_restb; w.S[k] = A.S; w.G[k] = G[i];
        w.lda0[k] = A.lda0; w.lda0b[k] = A.lda0b;
        w.n_dev[k] = n_dev[i]; w.N[k] = N[i];
        z += qt_cdiv(N[i], WGRAD_GROUP_ROWS);
        w.zend[k] = z;
        ++k;
    }
    if (k == 0) return QT_OK;
    for (int i = k; i < MAXSEG; ++i) {
        w.a0[i] = w.a_rest[i] = w.a0b[i] = w.a_restb[i] = w.S[i] = w.G[i] = nullptr;
        w.n_dev[i] = nullptr; w.N[i] = 0; w.zend[i] = z; w.lda0[i] = w.lda0b[i] = 0;
    }
    w.nseg = k; w.Ka = Ka; w.Ca = Ca; w.Cab = Cab; w.Ks = Ks; w.Co = Co; w.rows = WGRAD_GROUP_ROWS; w.part = part;
    w.Gn = Gn; w.ytiles = qt_cdiv(Co, BN); w.ldg = ldg; w.gpl = gpl; w.gsA = gsA; w.gsG = gsG; w.per_node = per_node;
    const int M = Ka * (Ca + Cab) + Ks;
    const dim3 grid(qt_cdiv(M, wgrad_fw(M) * 32), w.ytiles * Gn, z);
    QT_WGRAD_DISPATCH(k_gemm_wgrad_group, M, Co, grid, stream, w);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_wgrad_group(int nseg, const float* const* a0, const int* lda0, const float* const* a_rest,
                              const float* const* a0b, const int* lda0b, const float* const* a_restb, const float* const* S, const float* const* G, const int* N,
                              const int32_t* const* n_dev, int Ka, int Ca, int Cab, int Ks, int Co, float* part, int planes_sm,
                              void* stream) {
    QT_ARG(nseg >= 1 && nseg <= MAXSEG && a0 && G && N && n_dev && part, "1..16 uses per launch");
    QT_ARG(Ka >= 1 && Ca >= 1 && Cab >= 0 && Co >= 1 && (Ka == 1 || a_rest) && (Ks == 0 || S) && (Cab == 0 || (a0b && (Ka == 1 || a_restb))),
           "bad arguments");
    QT_ARG(Co % 4 == 0, "Co must be a multiple of 4 (float4 operands)");
    return wgrad_group_launch(__func__, nseg, a0, lda0, a_rest, a0b, lda0b, a_restb, S, G, N, n_dev, Ka, Ca, Cab, Ks, Co, Co, 0, 1, 0, 0, 0,
                              part, stream, planes_sm);
}

// qt_wgrad_group for the Gn weights of qt_proj_group at once: use s multiplies [A_g | S]^T (A_g = a0[s] + g gsA, Cin columns,
// row stride lda0[s]) with the gradient rows G[s] + g gsG (Co columns, row stride ldg; gpl > 0: in Co / gpl planes (N[s], gpl)); per_node: gsA / gsG count floats per node (x N[s]).
// part: (qt_wgrad_group_blocks(nseg, N), Gn, Cin + Ks, Co), overwritten; qt_colsum over the blocks gives the (Gn, Cin + Ks, Co) gradient.
extern "C" int qt_wgrad_groups(int nseg, const float* const* a0, const int* lda0, const float* const* S, const float* const* G,
                               const int* N, const int32_t* const* n_dev, int Cin, int Ks, int Co, int ldg, int gpl, int Gn,
                               int64_t gsA, int64_t gsG, int per_node, float* part, void* stream) {
    QT_ARG(nseg >= 1 && nseg <= MAXSEG && a0 && lda0 && G && N && n_dev && part, "1..16 uses per launch");
    QT_ARG(Cin >= 4 && Co >= 4 && Co % 4 == 0 && (Ks == 0 || S) && Gn >= 1 && ldg % 4 == 0 && gsA % 4 == 0 && gsG % 4 == 0 &&
           (gpl ? (gpl % 4 == 0 && Co % gpl == 0 && ldg >= gpl) : ldg >= Co), "bad arguments");
    QT_ARG((int64_t)qt_cdiv(Co, BN) * Gn <= 65535, "too many groups");
    return wgrad_group_launch(__func__, nseg, a0, lda0, nullptr, nullptr, nullptr, nullptr, S, G, N, n_dev, 1, Cin, 0, Ks, Co, ldg, gpl, Gn,
                              gsA, gsG, per_node, part, stream);
}

extern "C" int qt_colsum(const float* part, int nblk, int64_t len, float* out, void* stream) {
    QT_ARG(part && out && len > 0 && nblk >= 0, "bad arguments");
    hipLaunchKernelGGL(k_colsum, dim3(qt_cdiv(len, 32)), dim3(1024), 0, (hipStream_t)stream, part, nblk, len, out);
    QT_LAUNCHED();
    return QT_OK;
}
