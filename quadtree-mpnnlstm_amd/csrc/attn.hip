// Edge-softmax attention of TransformerConv(heads=1, concat=False, beta=False, edge_dim=2, root_weight=True), the
// convolution the reference's sea-ice scripts hard-code (model/model.py:51, ice_exp.py:48; PyG 2.2.0, restated in
// oracle/qt_oracle.py:transformer_conv).  SURVEY.md 8(f) row 1.
//
//   out_i = sum_{j -> i} d_ij alpha_ij (v_j + e_ij) + skip_i,   alpha_i. = softmax_j( q_i . (k_j + e_ij) / sqrt(C) )
//   e_ij  = We [angle(j -> i), dist(j, i)]          (lin_edge has no bias)
//
// q | k | v | skip are the four column blocks of ONE projection GEMM (proj, row stride ld).  Incoming edges of i are
// row i of the mesh CSR (the quadtree adjacency is symmetric) plus the self pair (i, i) that get_adj emits for
// multi-pixel cells (attrs (0, 0)); edge attributes are recomputed from the node centroids instead of being stored.
// One node per group of C/4 lanes (float4 each), online softmax, dot products reduced with xor shuffles.
// Backward in gather form, no atomics: pass A per target (dq_i + the messages' coefficients), pass B per source (dk_j, dv_j, dWe partials).
#include "qt_common.h"
#include <math.h>

namespace {

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
template <int LPN>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int d = 1; d < LPN; d <<= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float dot4(const F4& a, const F4& b) {
    return (a.v[0] * b.v[0] + a.v[1] * b.v[1]) + (a.v[2] * b.v[2] + a.v[3] * b.v[3]);
}

struct AttnArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* xy;        // (N, 2): centroid x, y in edge-attribute units
    const float* eattr;     // (E, 2) [angle, dist] of the message col[e] -> row(e), or NULL: recomputed from xy per lane
    const float* selfloop;  // (N) > 0 where the node carries a self pair, or NULL
    const float* proj;      // (N, ld): q | k | v | skip, C columns each
    const float* We;        // (C, 2)
    int ld, C, Ncap;
    const int32_t* n_dev;
    float scale;            // 1 / sqrt(real channel count)
    float keep;             // 1 - dropout p (1 = no dropout)
    const int32_t* rev;     // (E) position of the transposed entry (row col[e], column row(e)), with coef: see k_attn_bwd_source
    float* coef;            // (E + N, 2) backward scratch: (ds, alpha d) of the message whose transposed entry is stored at the slot, then of the self pairs
    int E;
    int ld_g;               // backward: row stride of g (a column block of a wider gradient is read in place)
    int accumulate;         // backward: add the We partials into `part` (several uses of one convolution share the slab)
    uint32_t seed;
    const uint32_t* seed_dev;   // optional step counter on the device, mixed into the seed: a captured launch (fixed `seed`
                                // argument) then still draws a new dropout mask at every replay
    // G convolutions on the same mesh in one launch (blockIdx.y = head): head g reads the column block [g 4C, (g+1) 4C) of the
    // proj rows (ld >= G 4C), We[g], writes the column block g C of the out rows (ld_o) and its own stats / Dn / coef / part
    // slabs.  gmod: head g reads the column block (g % gmod) C of the incoming gradient (a sum of head groups downstream).
    int ld_o, gmod;
    // strides in floats: ps between the q / k / v / skip blocks of a proj (and gproj) row, hs between heads of proj, hs_o of out,
    // hs_g of g.  Rows side by side (one (N, G 4C) matrix): ld = G 4C, ps = C, hs = 4C; one dense (N, C) plane per block and head:
    // ld = C, ps = N C, hs = 4 N C -- gathered k / v rows are then whole 128-byte lines of a contiguous array.
    int64_t ps, hs, hs_o, hs_g;
};

// per-head views of the operands (head = blockIdx.y; a single convolution is head 0 of 1)
#ifndef QT_ATTN_DESC
#define QT_ATTN_DESC 3      // bit 0 / 1 / 2: the forward / target / source pass walks the heads from the last to the first
#endif
template <int PASS>
__device__ __forceinline__ int head_setup(AttnArgs& a) {
#ifdef QT_ATTN_HEADS_ASCENDING
    const int hd = blockIdx.y;
#else
    if (!((QT_ATTN_DESC >> PASS) & 1)) {
        const int hd0 = blockIdx.y;
        if (hd0) {
            a.proj += hd0 * a.hs;
            a.We += (int64_t)hd0 * 2 * a.C;
            if (a.coef) a.coef += (int64_t)hd0 * 2 * ((int64_t)a.E + a.Ncap);
            a.seed += (uint32_t)hd0 * 0x632BE5ABu;
        }
        return hd0;
    }
    // Last in, first out over the heads: an 8-head operand is 510 MB, the memory-side cache 256 MB, so a launch should start with
    // the heads its predecessor touched last.  The grouped projection writes heads 0..7 -> the forward pass walks 7..0; in the
    // backward the target pass walks 7..0, the source pass 0..7 (it re-reads the q planes the target pass just read), and the data
    // gradient after it takes its groups 7..0 (qt_proj_group, reverse).  cfg4t: 47.1 ms per step with every launch ascending,
    // 46.6 with the three passes descending, 46.2 with this order.
    const int hd = (int)gridDim.y - 1 - (int)blockIdx.y;
#endif
    if (hd) {
        a.proj += hd * a.hs;
        a.We += (int64_t)hd * 2 * a.C;
        if (a.coef) a.coef += (int64_t)hd * 2 * ((int64_t)a.E + a.Ncap);
        a.seed += (uint32_t)hd * 0x632BE5ABu;        // every head draws its own attention-dropout mask
    }
    return hd;
}

__device__ __forceinline__ uint32_t eff_seed(const AttnArgs& a) {
    return a.seed_dev ? a.seed ^ (*a.seed_dev * 0x9E3779B9u) : a.seed;
}

// [angle, dist] of the message j -> i from (dx, dy) = xy_j - xy_i (graph_functions.py:358-370): atan2(dx, dy) mod 2pi / 2pi
__device__ __forceinline__ void edge_attr_xy(float dx, float dy, float* ang, float* dst) {
    float a = atan2f(dx, dy);
    if (a < 0.0f) a += 6.283185307179586f;
    *ang = a / 6.283185307179586f;
    *dst = sqrtf(dy * dy + dx * dx);
}

// inverted-dropout multiplier of the attention coefficient of edge (j -> i): the same in forward and both backward passes
__device__ __forceinline__ float drop_mult(uint32_t seed, int i, int j, float keep) {
    if (keep >= 1.0f) return 1.0f;
    uint32_t h = seed ^ ((uint32_t)i * 0x9E3779B1u) ^ ((uint32_t)j * 0x85EBCA77u);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return ((h >> 8) * (1.0f / 16777216.0f)) < keep ? 1.0f / keep : 0.0f;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with a private L2: XCD x gets the contiguous node range
// [x * chunk, (x + 1) * chunk) (chunk from the VALID rows), so a node's neighbours -- close in the mesh's node order -- sit in
// the L2 that gathers them (the same mapping as k_spmm).  Returns the node-group index of this workgroup or -1.
__device__ __forceinline__ int xcd_block(int rows, int nodes_per_block) {
    const int nblk = (rows + nodes_per_block - 1) / nodes_per_block;
    const int chunk = (nblk + 7) >> 3;
    const int bid = blockIdx.x;
    if ((bid >> 3) >= chunk) return -1;
    return (bid & 7) * chunk + (bid >> 3);
}

#ifndef QT_ATTN_EPT
#define QT_ATTN_EPT 2     // edges per trip (their gathers are issued together).  8 heads at the cfg4 shapes, forward / target /
                          // source pass: 1: 178 / 258 / 194 us, 2: 158 / 266 / 181, 4: 202 / 313 / 231 -- fewer registers, more waves
#endif
constexpr int EPT = QT_ATTN_EPT;
// register budgets: waves per SIMD the forward / target / source kernels are compiled for (0 = the compiler's choice)
#ifndef QT_ATTN_OCC_F
#define QT_ATTN_OCC_F 0
#endif
#ifndef QT_ATTN_OCC_T
#define QT_ATTN_OCC_T 0
#endif
#ifndef QT_ATTN_OCC_S
#define QT_ATTN_OCC_S 0
#endif
#define QT_WAVES_ATTR_(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#define QT_WAVES_ATTR(n) QT_WAVES_ATTR_(n)
#if QT_ATTN_OCC_F
#define QT_ATTN_WAVES_F QT_WAVES_ATTR(QT_ATTN_OCC_F)
#else
#define QT_ATTN_WAVES_F
#endif
#if QT_ATTN_OCC_T
#define QT_ATTN_WAVES_T QT_WAVES_ATTR(QT_ATTN_OCC_T)
#else
#define QT_ATTN_WAVES_T
#endif
#if QT_ATTN_OCC_S
#define QT_ATTN_WAVES_S QT_WAVES_ATTR(QT_ATTN_OCC_S)
#else
#define QT_ATTN_WAVES_S
#endif
#ifndef QT_ATTN_BS
#define QT_ATTN_BS 64      // one wave per workgroup (forward 32.4 -> 31.0 us at the cfg4 shapes; 128: 31.3)
#endif
// Addresses: the block bases (head, q / k / v / skip block) are uniform, so a row is reached as base + a 32-bit element offset
// (one VGPR per address instead of a 64-bit pair; the host checks that a head's rows span < 2^31 floats).
template <int LPN>
__global__ __launch_bounds__(QT_ATTN_BS) QT_ATTN_WAVES_F void k_attn_fwd(AttnArgs a, float* __restrict__ out, float* __restrict__ stats) {
    const int hd = head_setup<0>(a);
    out += hd * a.hs_o;
    stats += (int64_t)hd * 2 * a.Ncap;
    const int rows = qt_rows(a.n_dev, a.Ncap);
    const int blk = xcd_block(rows, QT_ATTN_BS / LPN);
    if (blk < 0) return;
    const int i = blk * (QT_ATTN_BS / LPN) + (int)threadIdx.x / LPN;
    if (i >= rows) return;
    const uint32_t j0 = ((uint32_t)threadIdx.x % LPN) * 4, ld = a.ld;
    const float* __restrict__ kb = a.proj + a.ps;
    const float* __restrict__ vb = a.proj + 2 * a.ps;
    const F4 q = ld4(a.proj + ((uint32_t)i * ld + j0));
    const F4 w0 = {{a.We[2 * j0], a.We[2 * j0 + 2], a.We[2 * j0 + 4], a.We[2 * j0 + 6]}};
    const F4 w1 = {{a.We[2 * j0 + 1], a.We[2 * j0 + 3], a.We[2 * j0 + 5], a.We[2 * j0 + 7]}};
    float m = -INFINITY, l = 0.0f;
    F4 acc = {{0, 0, 0, 0}};
    const int e0 = a.rowptr[i], e1 = a.rowptr[i + 1];
    const int extra = (a.selfloop && a.selfloop[i] > 0.0f) ? 1 : 0;
    const uint32_t seed = eff_seed(a);
    const int eend = e1 + extra;
    for (int eb = e0; eb < eend; eb += EPT) {
        // the gathers of EPT edges are issued together; the online-softmax updates then run in edge order
        int jj[EPT];
        F4 kk[EPT], vv[EPT];
        float2 ea[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = eb + u;
            jj[u] = e < e1 ? a.col[e] : (e < eend ? i : -1);
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (jj[u] >= 0) {
                const uint32_t off = (uint32_t)jj[u] * ld + j0;
                kk[u] = ld4(kb + off);
                vv[u] = ld4(vb + off);
                ea[u] = eb + u < e1 ? *reinterpret_cast<const float2*>(a.eattr + 2 * (uint32_t)(eb + u)) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (jj[u] < 0) break;                  // (uniform over the node's lane group)
            const int j = jj[u];
            const float ang = ea[u].x, dst = ea[u].y;       // stored per edge: no atan2 / sqrt in the loop
            F4 kj = kk[u], vj = vv[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ee = w0.v[c] * ang + w1.v[c] * dst;
                kj.v[c] += ee;
                vj.v[c] += ee;
            }
            const float s = group_sum<LPN>(dot4(q, kj)) * a.scale;
            const float mn = fmaxf(m, s);
            const float r = __expf(m - mn), p = __expf(s - mn);       // m = -inf on the first edge: r = 0
            const float pd = p * drop_mult(seed, i, j, a.keep);
            l = l * r + p;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc.v[c] = acc.v[c] * r + pd * vj.v[c];
            m = mn;
        }
    }
    const F4 sk = ld4(a.proj + 3 * a.ps + ((uint32_t)i * ld + j0));
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    F4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o.v[c] = acc.v[c] * inv + sk.v[c];
    st4(out + ((uint32_t)i * (uint32_t)a.ld_o + j0), o);
    if (j0 == 0) {
        stats[2 * i] = m;
        stats[2 * i + 1] = l;
    }
}

// pass A: per target i -- dq_i = scale * sum_e alpha_e (t_e - D_i)(k_j + e), t_e = d_e g_i.(v_j + e), D_i = sum_e alpha_e t_e.
// D_i needs no edge loop of its own: sum_e alpha_e d_e (v_j + e) is the forward's attention output out_i - skip_i, so
// D_i = g_i . (out_i - skip_i) is known BEFORE the loop, and the loop can hand every message's final coefficients
// (ds_e, ad_e) = (scale alpha_e (t_e - D_i), alpha_e d_e) to the SOURCE of the message: they are written at rev[e], the slot of the
// transposed entry (row j, column i), which pass B then reads in row order -- no gather of coefficients, softmax statistics or D_i
// over there, only the q_i / g_i rows.
template <int LPN>
__global__ __launch_bounds__(QT_ATTN_BS) QT_ATTN_WAVES_T void k_attn_bwd_target(AttnArgs a, const float* __restrict__ g, const float* __restrict__ stats,
                                                         const float* __restrict__ outf, float* __restrict__ gproj) {
    const int hd = head_setup<1>(a);
    g += (hd % a.gmod) * a.hs_g;
    stats += (int64_t)hd * 2 * a.Ncap;
    gproj += hd * a.hs;
    outf += hd * a.hs_o;
    const int rows = qt_rows(a.n_dev, a.Ncap);
    const int blk = xcd_block(rows, QT_ATTN_BS / LPN);
    if (blk < 0) return;
    const int i = blk * (QT_ATTN_BS / LPN) + (int)threadIdx.x / LPN;
    if (i >= rows) return;
    const uint32_t j0 = ((uint32_t)threadIdx.x % LPN) * 4, ld = a.ld;
    const uint32_t oi = (uint32_t)i * ld + j0;
    const float* __restrict__ kb = a.proj + a.ps;
    const float* __restrict__ vb = a.proj + 2 * a.ps;
    const F4 q = ld4(a.proj + oi), gi = ld4(g + ((uint32_t)i * (uint32_t)a.ld_g + j0));
    float D;
    {
        const F4 sk = ld4(a.proj + 3 * a.ps + oi), of = ld4(outf + ((uint32_t)i * (uint32_t)a.ld_o + j0));
        F4 att;
#pragma unroll
        for (int c = 0; c < 4; ++c) att.v[c] = of.v[c] - sk.v[c];
        D = group_sum<LPN>(dot4(gi, att));
    }
    const F4 w0 = {{a.We[2 * j0], a.We[2 * j0 + 2], a.We[2 * j0 + 4], a.We[2 * j0 + 6]}};
    const F4 w1 = {{a.We[2 * j0 + 1], a.We[2 * j0 + 3], a.We[2 * j0 + 5], a.We[2 * j0 + 7]}};
    const float m = stats[2 * i], l = stats[2 * i + 1];
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    const int e0 = a.rowptr[i], e1 = a.rowptr[i + 1];
    const int extra = (a.selfloop && a.selfloop[i] > 0.0f) ? 1 : 0;
    F4 dq = {{0, 0, 0, 0}};
    const uint32_t seed = eff_seed(a);
    const int eend = e1 + extra;
    for (int eb = e0; eb < eend; eb += EPT) {
        int jj[EPT], rv[EPT];
        F4 kk[EPT], vv[EPT];
        float2 ea[EPT];
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = eb + u;
            jj[u] = e < e1 ? a.col[e] : (e < eend ? i : -1);
            rv[u] = e < e1 ? a.rev[e] : a.E + i;
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u)
            if (jj[u] >= 0) {
                const uint32_t off = (uint32_t)jj[u] * ld + j0;
                kk[u] = ld4(kb + off);
                vv[u] = ld4(vb + off);
                ea[u] = eb + u < e1 ? *reinterpret_cast<const float2*>(a.eattr + 2 * (uint32_t)(eb + u)) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (jj[u] < 0) break;
            const int j = jj[u];
            const float ang = ea[u].x, dst = ea[u].y;
            F4 kj = kk[u], vj = vv[u];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ee = w0.v[c] * ang + w1.v[c] * dst;
                kj.v[c] += ee;
                vj.v[c] += ee;
            }
            const float s = group_sum<LPN>(dot4(q, kj)) * a.scale;
            const float alpha = __expf(s - m) * inv;
            const float d = drop_mult(seed, i, j, a.keep);
            const float t = d * group_sum<LPN>(dot4(gi, vj));
            const float ds = alpha * (t - D) * a.scale;
            if (j0 == 0) *reinterpret_cast<float2*>(a.coef + 2 * (uint32_t)rv[u]) = make_float2(ds, alpha * d);
#pragma unroll
            for (int c = 0; c < 4; ++c) dq.v[c] += ds * kj.v[c];
        }
    }
    st4(gproj + oi, dq);
    if (!(a.accumulate & 2)) st4(gproj + 3 * a.ps + oi, gi);          // skip branch: identity (bit 1: g IS that block already)
}

// pass B: per source j over its outgoing messages j -> i (i runs over row j: the adjacency is symmetric).  Row j stores i -> j;
// the coefficients of j -> i were left at this row's own slots by pass A:
//   dk_j = sum_i ds q_i,  dv_j = sum_i ad g_i,  dWe = sum_edges (dk + dv terms) [angle, dist]
template <int LPN>
__global__ __launch_bounds__(256) QT_ATTN_WAVES_S void k_attn_bwd_source(AttnArgs a, const float* __restrict__ g, float* __restrict__ gproj,
                                                         float* __restrict__ part) {
    __shared__ float sm[4 * LPN * 2 * 4];
    const int hd = head_setup<2>(a);
    g += (hd % a.gmod) * a.hs_g;
    gproj += hd * a.hs;
    const uint32_t lj = threadIdx.x % LPN, j0 = lj * 4, ld = a.ld, ldg = a.ld_g;
    const int N = qt_rows(a.n_dev, a.Ncap);
    float acc[2][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[0][c] = acc[1][c] = 0.0f;
    // XCD x sweeps the contiguous eighth [x * per, (x + 1) * per) of the nodes with its share of the workgroups
    const bool split = gridDim.x >= 8;                      // (a handful of workgroups: plain sweep)
    const int xcd = split ? blockIdx.x & 7 : 0, wg = split ? blockIdx.x >> 3 : blockIdx.x;
    const int nwg = split ? (gridDim.x + 7 - xcd) >> 3 : gridDim.x;
    const int per = split ? (N + 7) >> 3 : N;
    const int jlo = xcd * per, jhi = jlo + per < N ? jlo + per : N;
    const int stride = nwg * (256 / LPN);
    for (int j = jlo + wg * (256 / LPN) + (int)threadIdx.x / LPN; j < jhi; j += stride) {
        F4 dk = {{0, 0, 0, 0}}, dv = {{0, 0, 0, 0}};
        const int e0 = a.rowptr[j], e1 = a.rowptr[j + 1];
        const int extra = (a.selfloop && a.selfloop[j] > 0.0f) ? 1 : 0;
        const int eend = e1 + extra;
        for (int eb = e0; eb < eend; eb += EPT) {
            int ii[EPT];
            F4 qq[EPT], gg[EPT];
            float2 cf[EPT], ea[EPT];
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = eb + u;
                ii[u] = e < e1 ? a.col[e] : (e < eend ? j : -1);
            }
#pragma unroll
            for (int u = 0; u < EPT; ++u)
                if (ii[u] >= 0) {
                    const int e = eb + u;
                    const uint32_t slot = e < e1 ? e : a.E + j;
                    qq[u] = ld4(a.proj + ((uint32_t)ii[u] * ld + j0));
                    gg[u] = ld4(g + ((uint32_t)ii[u] * ldg + j0));
                    cf[u] = *reinterpret_cast<const float2*>(a.coef + 2 * slot);
                    ea[u] = e < e1 ? *reinterpret_cast<const float2*>(a.eattr + 2 * (uint32_t)e) : make_float2(0.0f, 0.0f);
                }
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                if (ii[u] < 0) break;
                float ang = 0.0f, dst = 0.0f;
                if (ii[u] != j) {                        // row j holds the attributes of i -> j: same distance, opposite direction
                    ang = ea[u].x + 0.5f;
                    if (ang >= 1.0f) ang -= 1.0f;
                    dst = ea[u].y;
                }
                const float ds = cf[u].x, ad = cf[u].y;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dkc = ds * qq[u].v[c], dvc = ad * gg[u].v[c];
                    dk.v[c] += dkc;
                    dv.v[c] += dvc;
                    acc[0][c] += (dkc + dvc) * ang;      // d We[:, 0]
                    acc[1][c] += (dkc + dvc) * dst;      // d We[:, 1]
                }
            }
        }
        const uint32_t oj = (uint32_t)j * ld + j0;
        st4(gproj + a.ps + oj, dk);
        st4(gproj + 2 * a.ps + oj, dv);
    }
    // block reduction of the We partials (same scheme as the LSTM parameter gradients): [2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = acc[k][c];
#pragma unroll
            for (int dd = LPN; dd < 64; dd <<= 1) v += __shfl_xor(v, dd, 64);
            acc[k][c] = v;
        }
    if (lane < LPN)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) sm[(wave * LPN + lane) * 8 + k * 4 + c] = acc[k][c];
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * a.C; idx += 256) {
        const int k = idx / a.C, ch = idx % a.C;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += sm[(w * LPN + (ch >> 2)) * 8 + k * 4 + (ch & 3)];
        float* pp = part + ((int64_t)blockIdx.x * gridDim.y + hd) * 2 * a.C + idx;     // (block, head)[k][channel]; the host transposes to (C, 2)
        *pp = (a.accumulate & 1) ? *pp + s : s;
    }
}

// [angle, dist] of every stored edge, once per mesh (the convolutions of a cell, forward and backward, and every time step on
// the same mesh reuse it)
__global__ void k_attn_edge_attrs(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                  const float* __restrict__ xy, int Ncap, const int32_t* __restrict__ n_dev,
                                  float* __restrict__ eattr, int32_t* __restrict__ rev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qt_rows(n_dev, Ncap)) return;
    const float xi = xy[2 * i], yi = xy[2 * i + 1];
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        const int j = col[e];
        float ang = 0.0f, dst = 0.0f;
        if (j != i) edge_attr_xy(xy[2 * j] - xi, xy[2 * j + 1] - yi, &ang, &dst);
        eattr[2 * e] = ang;
        eattr[2 * e + 1] = dst;
        if (rev) {                              // the adjacency is symmetric: (j, i) is in row j (itself when absent)
            int r = e;
            for (int f = rowptr[j]; f < rowptr[j + 1]; ++f)
                if (col[f] == i) {
                    r = f;
                    break;
                }
            rev[e] = r;
        }
    }
}

inline bool c_ok(int C) { return C == 4 || C == 8 || C == 16 || C == 32 || C == 64 || C == 128; }

}  // namespace

#define QT_ATTN_DISPATCH_BS(C, KERNEL, grid, BS, stream, ...)                                                               \
    switch ((C) / 4) {                                                                                               \
        case 1: hipLaunchKernelGGL(KERNEL<1>, dim3(grid), dim3(BS), 0, (hipStream_t)stream, __VA_ARGS__); break;    \
        case 2: hipLaunchKernelGGL(KERNEL<2>, dim3(grid), dim3(BS), 0, (hipStream_t)stream, __VA_ARGS__); break;    \
        case 4: hipLaunchKernelGGL(KERNEL<4>, dim3(grid), dim3(BS), 0, (hipStream_t)stream, __VA_ARGS__); break;    \
        case 8: hipLaunchKernelGGL(KERNEL<8>, dim3(grid), dim3(BS), 0, (hipStream_t)stream, __VA_ARGS__); break;    \
        case 16: hipLaunchKernelGGL(KERNEL<16>, dim3(grid), dim3(BS), 0, (hipStream_t)stream, __VA_ARGS__); break;  \
        default: hipLaunchKernelGGL(KERNEL<32>, dim3(grid), dim3(BS), 0, (hipStream_t)stream, __VA_ARGS__); break;  \
    }

#define QT_ATTN_DISPATCH(C, KERNEL, grid, stream, ...) QT_ATTN_DISPATCH_BS(C, KERNEL, grid, 256, stream, __VA_ARGS__)

static int fill_args(AttnArgs* a, const int32_t* rowptr, const int32_t* col, const float* xy, const float* eattr, const float* selfloop,
                     const float* proj, int ld, const float* We, int C, int c_real, int N, const int32_t* n_dev,
                     float keep, uint32_t seed, const uint32_t* seed_dev) {
    a->rowptr = rowptr; a->col = col; a->xy = xy; a->eattr = eattr; a->selfloop = selfloop; a->proj = proj; a->We = We;
    a->ld = ld; a->C = C; a->Ncap = N; a->n_dev = n_dev; a->scale = 1.0f / sqrtf((float)c_real); a->keep = keep; a->seed = seed; a->seed_dev = seed_dev;
    return 0;
}

extern "C" int qt_attn_blocks(int N, int C) {
    if (N <= 0 || !c_ok(C)) return 0;
    const int need = qt_cdiv((int64_t)N * (C / 4), 256);
#ifndef QT_ATTN_BLOCKS
#define QT_ATTN_BLOCKS 4096
#endif
    return need < QT_ATTN_BLOCKS ? need : QT_ATTN_BLOCKS;     // (512 left two waves per SIMD for a gather-latency-bound sweep)
}

extern "C" int qt_attn_edge_attrs(const int32_t* rowptr, const int32_t* col, const float* xy, int N, const int32_t* n_dev,
                                  float* eattr, int32_t* rev, void* stream) {
    QT_ARG(rowptr && col && xy && eattr, "null pointer");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_attn_edge_attrs, dim3(qt_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, xy, N, n_dev, eattr, rev);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_attn_fwd(const int32_t* rowptr, const int32_t* col, const float* xy, const float* eattr, const float* selfloop,
                           const float* proj, int ld, const float* We, int C, int c_real, int N, const int32_t* n_dev,
                           float keep, uint32_t seed, const uint32_t* seed_dev, float* out, float* stats, int G, int ld_o, int64_t ps, int64_t hs,
                           int64_t hs_o, void* stream) {
    QT_ARG(rowptr && col && eattr && proj && We && out && stats, "null pointer");
    if (G <= 0) G = 1;
    if (ps == 0) ps = C;
    if (hs == 0) hs = 4 * C;
    if (hs_o == 0) hs_o = C;
    if (ld_o == 0) ld_o = G * C;
    QT_ARG(c_ok(C) && ld >= C && ld % 4 == 0 && c_real >= 1 && c_real <= C, "bad channel count / row stride");
    QT_ARG(G <= 64 && ld_o >= C && ld_o % 4 == 0 && ps % 4 == 0 && hs % 4 == 0 && hs_o % 4 == 0, "bad head count / strides");
    QT_ARG((int64_t)N * ld < (1ll << 31) && (int64_t)N * ld_o < (1ll << 31), "a head's rows must span fewer than 2^31 floats");
    if (N <= 0) return QT_OK;
    AttnArgs a;
    fill_args(&a, rowptr, col, xy, eattr, selfloop, proj, ld, We, C, c_real, N, n_dev, keep, seed, seed_dev);
    a.ld_g = C; a.accumulate = 0; a.rev = nullptr; a.coef = nullptr; a.E = 0; a.ld_o = ld_o; a.gmod = G;
    a.ps = ps; a.hs = hs; a.hs_o = hs_o; a.hs_g = C;
    const int grid = (qt_cdiv((int64_t)N * (C / 4), QT_ATTN_BS) + 7) & ~7;      // whole rounds over the 8 XCDs (xcd_block)
    QT_ATTN_DISPATCH_BS(C, k_attn_fwd, dim3(grid, G), QT_ATTN_BS, stream, a, out, stats);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_attn_bwd(const int32_t* rowptr, const int32_t* col, const float* xy, const float* eattr, const float* selfloop,
                           const float* proj, int ld, const float* We, int C, int c_real, int N, const int32_t* n_dev,
                           float keep, uint32_t seed, const uint32_t* seed_dev, const float* g, int ld_g, const float* stats,
                           const float* out, int ld_o, float* gproj, float* part, int accumulate, const int32_t* rev, float* coef, int E,
                           int G, int gmod, int64_t ps, int64_t hs, int64_t hs_g, int64_t hs_o, void* stream) {
    QT_ARG(rowptr && col && xy && eattr && proj && We && g && stats && out && gproj && part && rev && coef, "null pointer");
    if (G <= 0) G = 1;
    if (gmod <= 0) gmod = G;
    if (ps == 0) ps = C;
    if (hs == 0) hs = 4 * C;
    if (hs_g == 0) hs_g = C;
    if (hs_o == 0) hs_o = C;
    if (ld_o == 0) ld_o = G * C;
    QT_ARG(c_ok(C) && ld >= C && ld % 4 == 0 && c_real >= 1 && c_real <= C, "bad channel count / row stride");
    if (ld_g == 0) ld_g = gmod * C;
    QT_ARG(G <= 64 && gmod <= G && ld_g >= C && ld_g % 4 == 0 && ((uintptr_t)g & 15) == 0 && ps % 4 == 0 && hs % 4 == 0 && hs_g % 4 == 0 &&
           ld_o >= C && ld_o % 4 == 0 && hs_o % 4 == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)coef & 7) == 0 && E >= 0,
           "bad head count / strides / alignment");
    QT_ARG((int64_t)N * ld < (1ll << 31) && (int64_t)N * ld_o < (1ll << 31) && (int64_t)N * ld_g < (1ll << 31) && (int64_t)E + N < (1ll << 30),
           "a head's rows must span fewer than 2^31 floats");
    if (N <= 0) return QT_OK;
    AttnArgs a;
    fill_args(&a, rowptr, col, xy, eattr, selfloop, proj, ld, We, C, c_real, N, n_dev, keep, seed, seed_dev);
    a.ld_g = ld_g; a.accumulate = accumulate; a.rev = rev; a.coef = coef; a.E = E; a.ld_o = ld_o; a.gmod = gmod;
    a.ps = ps; a.hs = hs; a.hs_o = hs_o; a.hs_g = hs_g;
    const int grid = (qt_cdiv((int64_t)N * (C / 4), QT_ATTN_BS) + 7) & ~7;
#if !defined(QT_EXP_ATTN_ONLY) || QT_EXP_ATTN_ONLY == 1       // (diagnostics builds time one pass alone)
    QT_ATTN_DISPATCH_BS(C, k_attn_bwd_target, dim3(grid, G), QT_ATTN_BS, stream, a, g, stats, out, gproj);
#endif
    const int gridB = qt_attn_blocks(N, C);
#if !defined(QT_EXP_ATTN_ONLY) || QT_EXP_ATTN_ONLY == 2
    QT_ATTN_DISPATCH(C, k_attn_bwd_source, dim3(gridB, G), stream, a, g, gproj, part);
#endif
    QT_LAUNCHED();
    return QT_OK;
}
