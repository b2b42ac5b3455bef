// Mesh <-> image transfers by label map: flatten / unflatten
// (model/graph_functions.py:391-419, 451-458) without the dense (N, P) mapping, the fused
// remesh transfer of seq2seq.py:440-442 + 474-477, and the masked squared error of
// mpnnlstm.py:243-246.
//
// k_pool: one workgroup per 64x64 image tile.  Each thread owns a 4x4 pixel block and
// builds the 2x2 / 4x4 sums in registers; 8x8 .. 64x64 sums go through a small LDS
// pyramid.  A quadtree leaf of level L is exactly one level-L pyramid entry, so every
// node value is written once, in a fixed order, with no atomics.
#include "qt_common.h"

namespace {

struct PoolArgs {
    const float* img;
    int S;
    int64_t img_clip_stride;   // floats between clips of img (S*P*C when dense)
    const float* src_val;      // source node values: one (N_src, C) matrix, or
    const float* part[8];      // up to 8 matrices side by side (row strides part_ld, widths as float4 prefix part_end)
    int part_ld[8], part_end[8], nparts;
    const int32_t* src_labels;
    const float* src_npix;
    int src_inv;
    int C;
    const int32_t* labels;
    const uint8_t* level;
    const float* npix;
    int mean;
    int B, n, m, N;
    float* out;
    int out_stride, out_coff;
    float* opart[8];         // mesh -> mesh: the result as up to 8 dense matrices side by side (widths as float4 prefix
    int opart_w[8], opart_end[8], noparts;   // opart_end), so that every consumer reads dense rows; 0: one matrix `out`
    int tiles_r, tiles_c;
    const int32_t* cell;     // (N, 4) r, c, size, clip: given -> nodes up to 4x4 pixels go through k_pool_nodes
    const int32_t* n_dev;
    int big_only;            // tile kernel: only nodes of 8x8 pixels and more
    const int32_t* direct;   // mesh -> mesh, optional: per destination node the source node under its ONE pixel (written
                             // by qt_quadtree_stage3 when the destination mesh was built from the source mesh, or the
                             // transposed index), -1 = multi-pixel node (general path); NULL: general path for all
};

template <int VEC>
struct Vec {
    float v[VEC];
};

// address of channel chunk `ch` (VEC floats) of source node `sl`
template <int VEC>
__device__ __forceinline__ const float* src_chunk(const PoolArgs& a, int64_t sl, int ch) {
    if (VEC == 1 || a.nparts == 0) return a.src_val + sl * a.C + ch * VEC;
    int s = 0;
    while (ch >= a.part_end[s]) ++s;
    return a.part[s] + sl * a.part_ld[s] + (ch - (s ? a.part_end[s - 1] : 0)) * 4;
}

// address of channel chunk `ch` of output node `node` (step s)
template <int VEC>
__device__ __forceinline__ float* dst_chunk(const PoolArgs& a, int s, int64_t node, int ch) {
    if (VEC == 1 || a.noparts == 0) return a.out + ((int64_t)s * a.N + node) * a.out_stride + a.out_coff + ch * VEC;
    int o = 0;
    while (ch >= a.opart_end[o]) ++o;
    return a.opart[o] + node * a.opart_w[o] + (ch - (o ? a.opart_end[o - 1] : 0)) * 4;
}

template <int VEC>
__device__ __forceinline__ Vec<VEC> vload(const float* p) {
    Vec<VEC> r;
    if constexpr (VEC == 4) {
        const float4 f = *reinterpret_cast<const float4*>(p);
        r.v[0] = f.x; r.v[1] = f.y; r.v[2] = f.z; r.v[3] = f.w;
    } else {
        r.v[0] = *p;
    }
    return r;
}

template <int VEC>
__device__ __forceinline__ void vstore(float* p, const Vec<VEC>& r, float scale) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) p[k] = r.v[k] * scale;
}

// (experiment switches: the node kernel takes nodes up to QT_NODE_MAX_Z pixels wide, the tile kernel those from level QT_TILE_MIN_LV)
#ifndef QT_TILE_MIN_LV
#define QT_TILE_MIN_LV 3
#endif
#ifndef QT_NODE_MAX_Z
#define QT_NODE_MAX_Z 4
#endif
template <int VEC>
__device__ __forceinline__ void tile_body(const PoolArgs& a, int bx, int by, int ny, float* pyr) {
    const int t = threadIdx.x;
    const int tiles = a.tiles_r * a.tiles_c;
    const int b = bx / tiles, tile = bx % tiles;
    const int R0 = (tile / a.tiles_c) * 64, C0 = (tile % a.tiles_c) * 64;
    const int br = t >> 4, bc = t & 15;
    const int64_t P = (int64_t)a.n * a.m;
    const int32_t* lab_img = a.labels + b * P;
    const uint8_t* lvl_img = a.level + b * P;

    int lab[16], slab[16];
    unsigned lv_pack[4] = {0, 0, 0, 0};
    float sscale[16];
    bool big = false;
    // big-only mode: a node of 8x8 pixels or more covers whole 4x4 blocks, so the block's first pixel tells whether any
    // of its 16 pixels is this kernel's business
    bool mine = true;
    if (a.big_only && QT_TILE_MIN_LV >= 3) {
        const int r = R0 + 4 * br, c = C0 + 4 * bc;
        mine = r < a.n && c < a.m && lvl_img[(int64_t)r * a.m + c] >= 3;
    }
    bool any_mine_t = false;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = R0 + 4 * br + i, c = C0 + 4 * bc + j;
            const int q = i * 4 + j;
            lab[q] = -1;
            slab[q] = -1;
            sscale[q] = 1.0f;
            if (mine && r < a.n && c < a.m) {
                const int64_t p = (int64_t)r * a.m + c;
                lab[q] = lab_img[p];
                const unsigned lv = lvl_img[p];
                lv_pack[i] |= lv << (8 * j);
                big |= (lv >= 3);
                any_mine_t |= (lv >= QT_TILE_MIN_LV);
                if (a.src_labels && (!a.big_only || lv >= QT_TILE_MIN_LV)) {
                    slab[q] = a.src_labels[b * P + p];
                    if (a.src_inv && slab[q] >= 0) sscale[q] = 1.0f / a.src_npix[slab[q]];
                }
                if (a.big_only && lv < QT_TILE_MIN_LV) lab[q] = -1;       // not this kernel's pixel: no load, no store
            }
        }
    const bool any_big = __syncthreads_or(big ? 1 : 0) != 0;
    if (QT_TILE_MIN_LV >= 3) {
        if (a.big_only && !any_big) return;
    } else if (a.big_only && __syncthreads_or(any_mine_t ? 1 : 0) == 0) {
        return;
    }

    const int nch = a.C / VEC;
    const int total = (a.src_labels ? 1 : a.S) * nch;
    for (int it = by; it < total; it += ny) {
        const int s = it / nch, ch = it % nch;
        Vec<VEC> val[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = i * 4 + j;
                const int r = R0 + 4 * br + i, c = C0 + 4 * bc + j;
                Vec<VEC> x;
#pragma unroll
                for (int k = 0; k < VEC; ++k) x.v[k] = 0.0f;
                if (a.src_labels) {
                    if (slab[q] >= 0) {
                        x = vload<VEC>(src_chunk<VEC>(a, slab[q], ch));
#pragma unroll
                        for (int k = 0; k < VEC; ++k) x.v[k] *= sscale[q];
                    }
                } else if (lab[q] >= 0) {
                    x = vload<VEC>(a.img + (int64_t)b * a.img_clip_stride + ((int64_t)s * P + (int64_t)r * a.m + c) * a.C + ch * VEC);
                }
                val[q] = x;
            }
        // level 0 leaves
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned lv = (lv_pack[q >> 2] >> (8 * (q & 3))) & 0xff;
            if (lab[q] >= 0 && lv == 0) vstore<VEC>(dst_chunk<VEC>(a, s, lab[q], ch), val[q], 1.0f);
        }
        // 2x2 and 4x4 sums in registers
        Vec<VEC> s2;
#pragma unroll
        for (int k = 0; k < VEC; ++k) s2.v[k] = 0.0f;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int q0 = (2 * i2) * 4 + 2 * j2;
                Vec<VEC> s1;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    s1.v[k] = (val[q0].v[k] + val[q0 + 1].v[k]) + (val[q0 + 4].v[k] + val[q0 + 5].v[k]);
                    s2.v[k] += s1.v[k];
                }
                const unsigned lv = (lv_pack[2 * i2] >> (8 * (2 * j2))) & 0xff;
                if (lab[q0] >= 0 && lv == 1)
                    vstore<VEC>(dst_chunk<VEC>(a, s, lab[q0], ch), s1, a.mean ? 1.0f / a.npix[lab[q0]] : 1.0f);
            }
        if (lab[0] >= 0 && (lv_pack[0] & 0xff) == 2)
            vstore<VEC>(dst_chunk<VEC>(a, s, lab[0], ch), s2, a.mean ? 1.0f / a.npix[lab[0]] : 1.0f);

        if (any_big) {
            // LDS pyramid: level 2 (16x16) -> 3 (8x8) -> 4 (4x4) -> 5 (2x2) -> 6 (1)
#pragma unroll
            for (int k = 0; k < VEC; ++k) pyr[t * VEC + k] = s2.v[k];
            __syncthreads();
            int src_off = 0, src_dim = 16, dst_off = 256;
            for (int L = 3; L <= 6; ++L) {
                const int dim = src_dim >> 1;
                if (t < dim * dim) {
                    const int i = t / dim, j = t % dim;
                    Vec<VEC> acc;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const float* sp = pyr + (src_off + (2 * i) * src_dim + 2 * j) * VEC + k;
                        acc.v[k] = (sp[0] + sp[VEC]) + (sp[src_dim * VEC] + sp[(src_dim + 1) * VEC]);
                        pyr[(dst_off + t) * VEC + k] = acc.v[k];
                    }
                    const int r = R0 + (i << L), c = C0 + (j << L);
                    if (r < a.n && c < a.m) {
                        const int64_t p = (int64_t)r * a.m + c;
                        const int lb = lab_img[p];
                        if (lb >= 0 && lvl_img[p] == L)
                            vstore<VEC>(dst_chunk<VEC>(a, s, lb, ch), acc, a.mean ? 1.0f / a.npix[lb] : 1.0f);
                    }
                }
                __syncthreads();
                src_off = dst_off;
                src_dim = dim;
                dst_off += dim * dim;
            }
        }
    }
}

#ifndef QT_POOL_OCC
#define QT_POOL_OCC 1
#endif
template <int VEC>
__global__ __launch_bounds__(256, QT_POOL_OCC) void k_pool(PoolArgs a) {
    __shared__ float pyr[(256 + 64 + 16 + 4 + 1) * VEC];
    tile_body<VEC>(a, blockIdx.x, blockIdx.y, gridDim.y, pyr);
}

// The loss of a rollout touches every output step once, after the last step: the per-step launches (squared error, per-node
// sum of the target for the gradient, the gradient itself) become one launch each over up to 16 steps, each step with its
// own mesh (pointer tables in the kernel arguments).
struct LossSeg {
    const float* out[16];         // node values of the step (column 0 of rows of out_stride floats)
    const int32_t* labels[16];
    const uint8_t* level[16];
    const float* npix[16];
    const int32_t* n_dev[16];
    float* sy[16];                // per-node sum of the target over the node's pixels
    float* gout[16];
    int out_stride[16], N[16];
};

__global__ __launch_bounds__(256) void k_pool_targets(PoolArgs a, LossSeg sg, int64_t y_step_stride) {
    __shared__ float pyr[256 + 64 + 16 + 4 + 1];
    const int z = blockIdx.z;
    a.img = a.img + z * y_step_stride;
    a.labels = sg.labels[z];
    a.level = sg.level[z];
    a.npix = sg.npix[z];
    a.N = sg.N[z];
    a.out = sg.sy[z];
    tile_body<1>(a, blockIdx.x, 0, 1, pyr);
}

// Node-centric transfer for nodes of 1x1 .. 4x4 pixels: thread = (node, float4 chunk of its row), chunk fastest, so the
// source-row gathers and the output stores move whole rows and consecutive threads write consecutive memory.  (The tile
// kernel above touches 16 bytes of every row per workgroup and has only B * tiles workgroups of pixel-serial work:
// 39 us for 68 channels at 64x64x32, against a ~16 us stream.)  Masked pixels inside a cell carry another label.
#ifndef QT_NODES_BS
#define QT_NODES_BS 256
#endif
#ifndef QT_POOL_CPT
#define QT_POOL_CPT 1      // row chunks per thread
#endif
// thread = (node, group of CPT consecutive VEC-float chunks of its row), group fastest.  One chunk per thread is the fastest:
// at the bench shape (68 channels, forward / backward transfer) 23.2 / 36.8 us with 1, 28.1 / 44.2 with 2, 37.1 / 66.6 with 4 --
// the waves that hold a 2x2 / 4x4 node set the launch time, and more chunks per thread lengthen exactly those.
#ifndef QT_POOL_CPT_A
#define QT_POOL_CPT_A 2    // row chunks per thread of the direct-copy role
#endif
// Direct-copy role of k_pool_nodes: destination nodes whose single source row is known (a.direct[i] >= 0: a single-pixel node of a
// mesh -> mesh transfer) are a gathered row copy -- index -> CPTA chunks of the row in flight -> stores, no cell record, no pixel
// loop.  They are nine tenths of a noisy frame's nodes; in their own workgroups they stream, instead of sharing waves with the few
// 2x2 / 4x4 nodes whose pixel loops set the time of every wave they sit in.
template <int VEC, int CPTA>
__device__ __forceinline__ void pool_direct_role(const PoolArgs& a, unsigned bid) {
    const int nch = a.C / VEC;
    const int ngrp = (nch + CPTA - 1) / CPTA;
    const unsigned idx = bid * QT_NODES_BS + threadIdx.x;
    const int64_t i = idx / (unsigned)ngrp;
    if (i >= qt_rows(a.n_dev, a.N)) return;
    const int g0 = (int)(idx - (unsigned)i * (unsigned)ngrp);      // this thread's chunks: g0, g0 + ngrp, ... (consecutive lanes take
    const int d = a.direct[i];                                      // consecutive chunks in every load instruction)
    if (d < 0) return;
    Vec<VEC> x[CPTA];
#pragma unroll
    for (int u = 0; u < CPTA; ++u)
        if (g0 + u * ngrp < nch) x[u] = vload<VEC>(src_chunk<VEC>(a, d, g0 + u * ngrp));
    const float sc = a.src_inv ? 1.0f / a.src_npix[d] : 1.0f;
#pragma unroll
    for (int u = 0; u < CPTA; ++u)
        if (g0 + u * ngrp < nch) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) x[u].v[k] *= sc;
            vstore<VEC>(dst_chunk<VEC>(a, 0, i, g0 + u * ngrp), x[u], 1.0f);
        }
}

template <int VEC, int CPT = (VEC == 4 ? QT_POOL_CPT : 1)>
__global__ __launch_bounds__(QT_NODES_BS) void k_pool_nodes(PoolArgs a, int grid_b) {
    // workgroups [0, grid_b): every node that is not a direct copy (started first: they hold the long pixel loops);
    // workgroups [grid_b, ..): the direct copies (grid_b == gridDim.x: no direct index, everything goes the general way)
    if ((int)blockIdx.x >= grid_b) {
        pool_direct_role<VEC, (VEC == 4 ? QT_POOL_CPT_A : 1)>(a, blockIdx.x - grid_b);
        return;
    }
    const bool split = grid_b < (int)gridDim.x;
    const int nch = a.C / VEC;
    const int ngrp = (nch + CPT - 1) / CPT;
    const int per = (a.src_labels ? 1 : a.S) * ngrp;
    // 32-bit thread index and division (pool_launch checks N * per < 2^31): the 64-bit division by a run-time value that
    // stood here is a ~100-instruction routine per thread
    const unsigned idx = blockIdx.x * QT_NODES_BS + threadIdx.x;
    const int64_t i = idx / (unsigned)per;
    if (i >= qt_rows(a.n_dev, a.N)) return;
    if (split && a.direct[i] >= 0) return;          // a direct copy: the other role's
    const int rem = (int)(idx - (unsigned)i * (unsigned)per);
    const int s = rem / ngrp, ch0 = (rem - s * ngrp) * CPT;
    Vec<VEC> acc[CPT];
#pragma unroll
    for (int u = 0; u < CPT; ++u)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[u].v[k] = 0.0f;
    auto store_all = [&](float scale) {
#pragma unroll
        for (int u = 0; u < CPT; ++u)
            if (ch0 + u < nch) vstore<VEC>(dst_chunk<VEC>(a, s, i, ch0 + u), acc[u], scale);
    };
    // the chunks of source row sl, scaled by 1 / its pixel count where the source holds sums, added into acc
    auto add_row = [&](int64_t sl) {
        Vec<VEC> x[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u)
            if (ch0 + u < nch) x[u] = vload<VEC>(src_chunk<VEC>(a, sl, ch0 + u));
        const float sc = a.src_inv ? 1.0f / a.src_npix[sl] : 1.0f;
#pragma unroll
        for (int u = 0; u < CPT; ++u)
            if (ch0 + u < nch)
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[u].v[k] += x[u].v[k] * sc;
    };
    if (a.direct) {
        // single-pixel destination node whose source node is known: index -> row, two dependent loads instead of the
        // cell -> pixel label -> row chain (and 4 bytes of index per thread instead of the 16-byte cell record: the index
        // traffic was as large as the payload).  Same arithmetic as the single-pixel branch below: bit-identical rows.
        const int d = a.direct[i];
        if (d >= 0) {
            add_row(d);
            store_all(1.0f);
            return;
        }
    }
    const int4 cl = reinterpret_cast<const int4*>(a.cell)[i];
    if (cl.z > QT_NODE_MAX_Z) return;
    const int64_t P = (int64_t)a.n * a.m, base = (int64_t)cl.w * P;
    if (cl.z == 1 && a.src_labels) {
        // a single-pixel node (almost all of them on noisy frames) owns its pixel: straight-line code, no pixel loop, no
        // label check, and its pixel count is 1 (no npix load for the mean)
        const int sl = a.src_labels[base + (int64_t)cl.x * a.m + cl.y];
        if (sl >= 0) add_row(sl);
        store_all(1.0f);
        return;
    }
    const float oscale = a.mean ? 1.0f / a.npix[i] : 1.0f;      // (requested before the pixel loop, not after it)
    if (a.src_labels) {
        // 2x2 and 4x4 nodes, mesh -> mesh: FOUR pixels per trip, their label loads together and then their row gathers
        // together (two dependent memory phases per trip).  The pixel-by-pixel loop that stood here walked three dependent
        // loads per pixel -- 12 phases for a 2x2 node, 48 for a 4x4 one -- and although such nodes are a few per cent of a
        // noisy frame's mesh, a quarter of the waves holds one: those waves were the tail that set the launch time.
        const int z = cl.z;                           // 2 or 4 (3 never occurs: cells are powers of two)
        // the labels of ALL the node's pixels first (one memory phase), then the row gathers four at a time in pixel order: 1 + 1
        // dependent phases for a 2x2 node, 1 + 4 for a 4x4 one (label and gather phases alternated before: 2 and 8)
        int sl[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            sl[q] = -1;
            if (q < z * z) {
                const int r = cl.x + (z == 2 ? (q >> 1) : (q >> 2)), c = cl.y + (z == 2 ? (q & 1) : (q & 3));
                if (r < a.n && c < a.m) {
                    const int64_t p = base + (int64_t)r * a.m + c;
                    const int own = a.labels[p], s_ = a.src_labels[p];       // independent loads
                    sl[q] = own == (int)i ? s_ : -1;
                }
            }
        }
#pragma unroll
        for (int q0 = 0; q0 < 16; q0 += 4) {
            if (q0 >= z * z) break;
            Vec<VEC> x[4][CPT];
            float sc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int u = 0; u < CPT; ++u)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) x[q][u].v[k] = 0.0f;
                sc[q] = 1.0f;
                if (sl[q0 + q] >= 0) {
#pragma unroll
                    for (int u = 0; u < CPT; ++u)
                        if (ch0 + u < nch) x[q][u] = vload<VEC>(src_chunk<VEC>(a, sl[q0 + q], ch0 + u));
                    if (a.src_inv) sc[q] = 1.0f / a.src_npix[sl[q0 + q]];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < CPT; ++u)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[u].v[k] += x[q][u].v[k] * sc[q];
        }
        store_all(oscale);
        return;
    }
    for (int dr = 0; dr < cl.z; ++dr)
        for (int dc = 0; dc < cl.z; ++dc) {
            const int r = cl.x + dr, c = cl.y + dc;
            if (r >= a.n || c >= a.m) continue;
            const int64_t p = (int64_t)r * a.m + c;
            if (cl.z > 1 && a.labels[base + p] != (int)i) continue;      // (a single-pixel node owns its pixel)
#pragma unroll
            for (int u = 0; u < CPT; ++u)
                if (ch0 + u < nch) {
                    const Vec<VEC> x = vload<VEC>(a.img + (int64_t)cl.w * a.img_clip_stride + ((int64_t)s * P + p) * a.C + (ch0 + u) * VEC);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[u].v[k] += x.v[k];
                }
        }
    store_all(oscale);
}

template <int VEC>
__global__ void k_gather(const float* __restrict__ val, int C, const int32_t* __restrict__ labels,
                         const float* __restrict__ inv_npix, int64_t total, float* __restrict__ img) {
    const int nch = C / VEC;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total * nch) return;
    int64_t p;
    int ch;
    if (total * nch < ((int64_t)1 << 31)) {          // 32-bit division whenever the index fits (uniform branch)
        p = (unsigned)idx / (unsigned)nch;
        ch = (int)((unsigned)idx - (unsigned)p * (unsigned)nch);
    } else {
        p = idx / nch;
        ch = (int)(idx % nch);
    }
    const int lab = labels[p];
    Vec<VEC> x;
#pragma unroll
    for (int k = 0; k < VEC; ++k) x.v[k] = 0.0f;
    float sc = 1.0f;
    if (lab >= 0) {
        x = vload<VEC>(val + (int64_t)lab * C + ch * VEC);
        if (inv_npix) sc = 1.0f / inv_npix[lab];
    }
    vstore<VEC>(img + p * C + ch * VEC, x, sc);
}

__global__ __launch_bounds__(256) void k_sse(const float* __restrict__ out, int out_stride, const int32_t* __restrict__ labels,
                                             const float* __restrict__ y, int64_t y_clip_stride, int64_t P,
                                             float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t p = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (p < P) {
            const int lab = labels[b * P + p];
            if (lab >= 0) {
                const float d = out[(int64_t)lab * out_stride] - y[b * y_clip_stride + p];
                acc += d * d;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_sse_multi(LossSeg sg, const float* __restrict__ y, int64_t y_clip_stride,
                                                   int64_t y_step_stride, int64_t P, int B, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y, z = blockIdx.z;
    const float* out = sg.out[z];
    const int32_t* labels = sg.labels[z];
    const int os = sg.out_stride[z];
    const float* yz = y + z * y_step_stride + b * y_clip_stride;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t p = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (p < P) {
            const int lab = labels[b * P + p];
            if (lab >= 0) {
                const float d = out[(int64_t)lab * os] - yz[p];
                acc += d * d;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        partial[((int64_t)z * B + b) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void k_sse_bwd_multi(LossSeg sg, const float* __restrict__ g, int W) {
    const int z = blockIdx.y;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = idx / (unsigned)W;
    if (i >= qt_rows(sg.n_dev[z], sg.N[z])) return;
    const float* out = sg.out[z];
    sg.gout[z][idx] = (idx - (unsigned)i * (unsigned)W) == 0
                          ? 2.0f * g[0] * (sg.npix[z][i] * out[i * sg.out_stride[z]] - sg.sy[z][i]) : 0.0f;
}

__global__ void k_sse_bwd(const float* __restrict__ out, int out_stride, const float* __restrict__ npix,
                          const float* __restrict__ sy, const float* __restrict__ g, int Ncap, const int32_t* __restrict__ n_dev,
                          int W, float* __restrict__ gout) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;          // (N * W < 2^31 checked by the launcher)
    const int64_t i = idx / (unsigned)W;
    if (i >= qt_rows(n_dev, Ncap)) return;
    gout[idx] = (idx - (unsigned)i * (unsigned)W) == 0 ? 2.0f * g[0] * (npix[i] * out[i * out_stride] - sy[i]) : 0.0f;
}

}  // namespace

extern "C" int qt_sse_bwd(const float* out, int out_stride, const float* npix, const float* sy, const float* g, int N,
                          const int32_t* n_dev, int W, float* gout, void* stream) {
    QT_ARG(out && npix && sy && g && gout && W >= 1 && out_stride >= 1, "bad arguments");
    QT_ARG((int64_t)N * W + 256 < ((int64_t)1 << 31), "N * W too large for 32-bit thread indices");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_sse_bwd, dim3(qt_cdiv((int64_t)N * W, 256)), dim3(256), 0, (hipStream_t)stream, out, out_stride, npix,
                       sy, g, N, n_dev, W, gout);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_sse_rollout(int nseg, const float* const* outs, const int* out_strides, const int32_t* const* labels,
                              const uint8_t* const* levels, const int* Ns, float* const* sys, const float* y,
                              int64_t y_clip_stride, int64_t y_step_stride, int B, int n, int m, float* partial, void* stream) {
    QT_ARG(nseg >= 1 && nseg <= 16 && outs && out_strides && labels && levels && Ns && sys && y && partial && B > 0, "bad arguments");
    LossSeg sg = {};
    for (int z = 0; z < nseg; ++z) {
        QT_ARG(outs[z] && labels[z] && levels[z] && sys[z] && out_strides[z] >= 1, "null segment pointer");
        sg.out[z] = outs[z]; sg.out_stride[z] = out_strides[z]; sg.labels[z] = labels[z]; sg.level[z] = levels[z];
        sg.N[z] = Ns[z]; sg.sy[z] = sys[z];
    }
    const int64_t P = (int64_t)n * m;
    hipLaunchKernelGGL(k_sse_multi, dim3(qt_cdiv(P, 1024), B, nseg), dim3(256), 0, (hipStream_t)stream, sg, y, y_clip_stride,
                       y_step_stride, P, B, partial);
    QT_LAUNCHED();
    PoolArgs a = {};
    a.img = y; a.S = 1; a.img_clip_stride = y_clip_stride; a.C = 1; a.mean = 0; a.B = B; a.n = n; a.m = m;
    a.out_stride = 1; a.out_coff = 0; a.tiles_r = qt_cdiv(n, 64); a.tiles_c = qt_cdiv(m, 64);
    hipLaunchKernelGGL(k_pool_targets, dim3(B * a.tiles_r * a.tiles_c, 1, nseg), dim3(256), 0, (hipStream_t)stream, a, sg,
                       y_step_stride);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_sse_rollout_bwd(int nseg, const float* const* outs, const int* out_strides, const float* const* npixs,
                                  const float* const* sys, const int* Ns, const int32_t* const* n_devs, const float* g, int W,
                                  float* const* gouts, void* stream) {
    QT_ARG(nseg >= 1 && nseg <= 16 && outs && out_strides && npixs && sys && Ns && n_devs && g && gouts && W >= 1, "bad arguments");
    LossSeg sg = {};
    int nmax = 0;
    for (int z = 0; z < nseg; ++z) {
        QT_ARG(outs[z] && npixs[z] && sys[z] && gouts[z] && out_strides[z] >= 1, "null segment pointer");
        sg.out[z] = outs[z]; sg.out_stride[z] = out_strides[z]; sg.npix[z] = npixs[z]; sg.sy[z] = (float*)sys[z];
        sg.N[z] = Ns[z]; sg.n_dev[z] = n_devs[z]; sg.gout[z] = gouts[z];
        nmax = Ns[z] > nmax ? Ns[z] : nmax;
    }
    QT_ARG((int64_t)nmax * W + 256 < ((int64_t)1 << 31), "N * W too large for 32-bit thread indices");
    if (nmax <= 0) return QT_OK;
    hipLaunchKernelGGL(k_sse_bwd_multi, dim3(qt_cdiv((int64_t)nmax * W, 256), nseg), dim3(256), 0, (hipStream_t)stream, sg, g, W);
    QT_LAUNCHED();
    return QT_OK;
}

static int pool_launch(PoolArgs& a, bool v4, const int32_t* cell, const int32_t* n_dev, hipStream_t stream) {
    const int total = a.S * (v4 ? a.C / 4 : a.C);
    a.cell = cell; a.n_dev = n_dev; a.big_only = cell != nullptr;
    if (cell) {
        if ((int64_t)a.N * total + QT_NODES_BS >= ((int64_t)1 << 31)) {
            qt_set_error("%s: N * channels too large for 32-bit thread indices", __func__);
            return QT_E_ARG;
        }
        const int cpt = v4 ? QT_POOL_CPT : 1, nch = v4 ? a.C / 4 : a.C;
        const int grid_b = qt_cdiv((int64_t)a.N * (a.src_labels ? 1 : a.S) * ((nch + cpt - 1) / cpt), QT_NODES_BS);
        static const bool no_split = getenv("QT_POOL_NO_SPLIT") != nullptr;        // (diagnostics)
        const int cpta = v4 ? QT_POOL_CPT_A : 1;
        const int grid_a = (a.direct && a.src_labels && !no_split) ? qt_cdiv((int64_t)a.N * ((nch + cpta - 1) / cpta), QT_NODES_BS) : 0;
        if (v4)
            hipLaunchKernelGGL(k_pool_nodes<4>, dim3(grid_b + grid_a), dim3(QT_NODES_BS), 0, stream, a, grid_b);
        else
            hipLaunchKernelGGL(k_pool_nodes<1>, dim3(grid_b + grid_a), dim3(QT_NODES_BS), 0, stream, a, grid_b);
        QT_LAUNCHED();
    }
    const int blocks = a.B * a.tiles_r * a.tiles_c;
    int gy = total;
    if (blocks * gy > 2048) gy = max(1, 2048 / blocks);
    if (gy > total) gy = total;
    if (v4)
        hipLaunchKernelGGL(k_pool<4>, dim3(blocks, gy), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(k_pool<1>, dim3(blocks, gy), dim3(256), 0, stream, a);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_gather(const float* val, int C, const int32_t* labels, const float* inv_npix, int64_t npixels_total,
                         float* img, void* stream) {
    QT_ARG(val && labels && img && C > 0, "bad arguments");
    if (npixels_total <= 0) return QT_OK;
    const bool v4 = (C % 4 == 0) && (((uintptr_t)val | (uintptr_t)img) % 16 == 0);
    const int nch = v4 ? C / 4 : C;
    const int grid = qt_cdiv(npixels_total * nch, 256);
    if (v4)
        hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, val, C, labels, inv_npix, npixels_total, img);
    else
        hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, val, C, labels, inv_npix, npixels_total, img);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_pool(const float* img, int S, int64_t img_clip_stride, const float* src_val, const int32_t* src_labels, const float* src_npix,
                       int src_inv, int C, const int32_t* labels, const uint8_t* level, const float* npix, int mean,
                       int B, int n, int m, int N, const int32_t* cell, const int32_t* n_dev, float* out, int out_stride,
                       int out_coff, void* stream) {
    QT_ARG(labels && level && out && C > 0 && B > 0, "bad arguments");
    QT_ARG((img != nullptr) != (src_val != nullptr && src_labels != nullptr), "give either img or src_val+src_labels");
    QT_ARG(!mean || npix, "mean pooling needs npix");
    QT_ARG(!src_inv || src_npix, "src_inv needs src_npix");
    QT_ARG(out_stride >= out_coff + C, "output row too short");
    if (N <= 0) return QT_OK;
    PoolArgs a = {};
    a.img = img; a.S = img ? S : 1; a.img_clip_stride = img_clip_stride > 0 ? img_clip_stride : (int64_t)a.S * n * m * C; a.src_val = src_val; a.src_labels = src_labels; a.src_npix = src_npix;
    a.src_inv = src_inv; a.C = C; a.labels = labels; a.level = level; a.npix = npix; a.mean = mean;
    a.B = B; a.n = n; a.m = m; a.N = N; a.out = out; a.out_stride = out_stride; a.out_coff = out_coff;
    a.tiles_r = qt_cdiv(n, 64); a.tiles_c = qt_cdiv(m, 64);
    a.nparts = 0;
    const float* srcp = img ? img : src_val;
    const bool v4 = (C % 4 == 0) && ((uintptr_t)srcp % 16 == 0) && (img_clip_stride % 4 == 0);
    return pool_launch(a, v4, cell, n_dev, (hipStream_t)stream);
}

extern "C" int qt_remesh(const float* const* src_parts, const int* widths, const int* lds, int nparts,
                         const int32_t* src_labels, const float* src_npix, int src_inv, const int32_t* labels,
                         const uint8_t* level, const float* npix, int mean, int B, int n, int m, int N, const int32_t* cell,
                         const int32_t* n_dev, float* const* out_parts, const int* out_widths, int nout, const int32_t* direct,
                         void* stream) {
    QT_ARG(src_parts && widths && lds && nparts >= 1 && nparts <= 8 && src_labels && labels && level && B > 0, "bad arguments");
    QT_ARG(!direct || cell, "the direct source index serves the node kernel (needs cell)");
    QT_ARG(out_parts && out_widths && nout >= 1 && nout <= 8, "bad output parts");
    QT_ARG(!mean || npix, "mean pooling needs npix");
    QT_ARG(!src_inv || src_npix, "src_inv needs src_npix");
    PoolArgs a = {};
    int c4 = 0;
    for (int i = 0; i < nparts; ++i) {
        QT_ARG(src_parts[i] && widths[i] > 0 && widths[i] % 4 == 0 && lds[i] % 4 == 0 && lds[i] >= widths[i] &&
               ((uintptr_t)src_parts[i] & 15) == 0, "source parts must be 16-byte aligned with widths / strides that are multiples of 4");
        a.part[i] = src_parts[i];
        a.part_ld[i] = lds[i];
        c4 += widths[i] / 4;
        a.part_end[i] = c4;
    }
    for (int i = nparts; i < 8; ++i) { a.part[i] = nullptr; a.part_ld[i] = 0; a.part_end[i] = c4; }
    a.nparts = nparts;
    if (N <= 0) return QT_OK;
    a.img = nullptr; a.S = 1; a.img_clip_stride = 0; a.src_val = src_parts[0]; a.src_labels = src_labels; a.src_npix = src_npix;
    a.src_inv = src_inv; a.C = 4 * c4; a.labels = labels; a.level = level; a.npix = npix; a.mean = mean;
    int o4 = 0;
    for (int i = 0; i < nout; ++i) {
        QT_ARG(out_parts[i] && out_widths[i] > 0 && out_widths[i] % 4 == 0 && ((uintptr_t)out_parts[i] & 15) == 0,
               "output parts must be 16-byte aligned with widths that are multiples of 4");
        a.opart[i] = out_parts[i];
        a.opart_w[i] = out_widths[i];
        o4 += out_widths[i] / 4;
        a.opart_end[i] = o4;
    }
    for (int i = nout; i < 8; ++i) { a.opart[i] = nullptr; a.opart_w[i] = 0; a.opart_end[i] = o4; }
    QT_ARG(o4 == c4, "the output parts must add up to the source width");
    a.noparts = nout > 1 ? nout : 0;
    a.B = B; a.n = n; a.m = m; a.N = N; a.out = out_parts[0]; a.out_stride = 4 * c4; a.out_coff = 0;
    a.tiles_r = qt_cdiv(n, 64); a.tiles_c = qt_cdiv(m, 64);
    a.direct = direct;
    return pool_launch(a, true, cell, n_dev, (hipStream_t)stream);
}

extern "C" int qt_sse(const float* out, int out_stride, const int32_t* labels, const float* y, int64_t y_clip_stride,
                      int B, int n, int m, float* partial, void* stream) {
    QT_ARG(out && labels && y && partial, "null pointer");
    const int64_t P = (int64_t)n * m;
    hipLaunchKernelGGL(k_sse, dim3(qt_cdiv(P, 1024), B), dim3(256), 0, (hipStream_t)stream, out, out_stride, labels, y,
                       y_clip_stride, P, partial);
    QT_LAUNCHED();
    return QT_OK;
}
