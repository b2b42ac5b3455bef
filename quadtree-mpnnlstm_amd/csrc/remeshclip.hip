// Clip-resident mesh -> mesh state transfer (the re-mesh of model/seq2seq.py:434-491: unflatten on the old mesh + flatten on
// the new one, :440-442 + :474-477, fused; its backward is the same op with the meshes swapped).
//
//   out[new node i] = (mean ? 1 / npix[i] : 1) * sum over the pixels p of node i of  S[src_label[p]] (* 1 / src_npix[..] if src_inv)
//
// The general kernels (transfer.hip: k_pool_nodes for nodes up to 4 x 4 pixels, k_pool for the bigger ones) walk
// index -> label -> row chains through L2 / HBM for every node and channel chunk, 41 - 52 us per transfer of the benchmark's
// 68 state channels.  The transfer of one 64 x 64 tile (= base cell) of a clip and one 4-channel column slice fits a
// workgroup's LDS -- a base cell's nodes are one contiguous label range of the DFS order, so its source rows are a range too
// (frames of several tiles: both meshes decomposed with max_size 64; frames up to 64 x 64: any mesh, the clip is the tile):
//   * the source slice of the clip (<= 4096 rows x 16 B = 64 KB) is staged once (pre-scaled by 1 / src_npix for the backward);
//   * every thread owns one 2 x 2 pixel block (Morton order): pixel value = staged row of the pixel's source node, an LDS gather;
//   * a sum pyramid over the 64 x 64 frame (level-1 sums in registers, levels 2 .. 6 through 22 KB of LDS) -- a quadtree leaf
//     of level L is exactly one level-L entry, so every destination node is written once, deterministically, by the thread
//     that owns its head pixel.
// No per-node cell record, no direct-index side array, no pixel loops of data-dependent length, no atomics.  Summation order:
// (top-left + top-right) + (bottom-left + bottom-right) at every level.
#include "qt_common.h"

namespace {

constexpr int RC_T = 1024;          // threads = 2 x 2 pixel blocks of a 64 x 64 frame
constexpr int RC_ROWS = 4096;       // source rows of a clip that fit

struct RcArgs {
    const float* part[8];           // source state as up to 8 matrices side by side (row strides part_ld, float4 prefix part_end)
    int part_ld[8], part_end[8];
    float* opart[8];                // the result likewise, dense rows (widths opart_w, float4 prefix opart_end)
    int opart_w[8], opart_end[8];
    const int32_t* src_labels;      // (B, n, m) source mesh
    const float* src_npix;
    const int32_t* src_off;         // (B * tiles + 1) first source node of every 64 x 64 tile, label order (slot b*T + (T-1-tile))
    const int32_t* labels;          // (B, n, m) destination mesh
    const uint8_t* level;
    const float* npix;
    int src_inv, mean, B, n, m, tiles_c, tiles;
    // the decoder's next input assembled by the transfer itself (model/seq2seq.py:484-487: [transferred output value | position,
    // size]): float4 chunk 0 of the result is written as (value.x, posfeat[node]) when posfeat != NULL -- and in the transposed
    // transfer only column 0 of chunk 0 of the SOURCE counts (src_first_only): the gradient of that assembly
    const float* posfeat;
    int src_first_only;
    int ride;                       // 1: chunk 0 is the decoder input's scalar (posfeat or src_first_only, and more chunks follow): it has no
                                    // workgroups of its own, the workgroups of chunk 1 carry its column 0 along
};

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ unsigned compact_bits(unsigned v) {       // even bits of v -> low half
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0F0F0F0Fu;
    v = (v | (v >> 4)) & 0x00FF00FFu;
    return v;
}
// Workgroup barrier that waits for this wave's LDS traffic only (not for its global loads / stores)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pairwise sums over lane groups of four, in the pyramid's order: lane 4j holds child 0 .. lane 4j+3 child 3 of entry j (Morton
// order, child = row bit | col bit << 1) -> every lane of the group gets (c0 + c2) + (c1 + c3) (a + b = b + a bit for bit, so
// the four lanes hold the same bits).  sh = log2 of the distance between the children: 0, 2, 4 for levels 2, 3, 4.
__device__ __forceinline__ float quad_sum(float v, int sh) {
    const float a = v + __shfl_xor(v, 2 << sh, 64);
    return a + __shfl_xor(a, 1 << sh, 64);
}
__device__ __forceinline__ float4 quad_sum4(float4 v, int sh) {
    return make_float4(quad_sum(v.x, sh), quad_sum(v.y, sh), quad_sum(v.z, sh), quad_sum(v.w, sh));
}

// 64 KB + 16 KB of LDS and <= 64 registers: TWO workgroups per CU.  The first version kept the whole pyramid in LDS (86 KB, one
// workgroup per CU, seven barriers) and the benchmark's 17 chunks x 32 clips = 544 workgroups ran in three rounds over the 256
// CUs; now levels 1 .. 4 are lane-group sums in registers (a wave's 64 threads are exactly one level-4 entry), levels 5 and 6
// are formed from the sixteen level-4 entries by the few threads that need them, and the decoder input's scalar chunk RIDES with
// chunk 1 (ride): 16 x 32 = 512 workgroups, all resident at once.
__global__ __launch_bounds__(RC_T, 2) void k_remesh_clip(RcArgs a) {
    __shared__ float4 A[RC_ROWS];                 // the clip's source slice; after the gathers its first 16 entries = the level-4 sums
    __shared__ float A0[RC_ROWS];                 // the rider: column 0 of chunk 0; likewise
    const int t = threadIdx.x;
    const int b = (int)blockIdx.x % a.B;
    const int rest = (int)blockIdx.x / a.B;
    const int tile = rest % a.tiles;
    const int ch = rest / a.tiles + a.ride;
    const bool ride = a.ride && ch == 1;          // (workgroup-uniform) this workgroup also transfers column 0 of chunk 0
    const int R0 = (tile / a.tiles_c) * 64, C0 = (tile % a.tiles_c) * 64;
    const int slot = b * a.tiles + (a.tiles - 1 - tile);
    int sp = 0, op = 0;
    while (ch >= a.part_end[sp]) ++sp;
    while (ch >= a.opart_end[op]) ++op;
    const float* src = a.part[sp] + (ch - (sp ? a.part_end[sp - 1] : 0)) * 4;
    const int lds = a.part_ld[sp];
    float* dst = a.opart[op] + (ch - (op ? a.opart_end[op - 1] : 0)) * 4;
    const int ldd = a.opart_w[op];
    const int r0s = a.src_off[slot];
    const int nrs = min(a.src_off[slot + 1] - r0s, RC_ROWS);
    const int P = a.n * a.m;

    // ---- one memory phase: the source slice (4 rows per thread) and the labels / levels of this thread's 2 x 2 pixels
    float4 x[4];
    float x0[4], sc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int lr = min(t + RC_T * u, nrs - 1);
        x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        x0[u] = 0.0f;
        sc[u] = 1.0f;
        if (nrs > 0) {
            x[u] = *reinterpret_cast<const float4*>(src + (int64_t)(r0s + lr) * lds);
            if (ride) x0[u] = a.part[0][(int64_t)(r0s + lr) * a.part_ld[0]];
            if (a.src_inv) sc[u] = a.src_npix[r0s + lr];
        }
    }
    const int br = (int)compact_bits((unsigned)t), bc = (int)compact_bits((unsigned)t >> 1);     // t = spread(br) | spread(bc) << 1
    int lab[4], sl[4], lv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = R0 + 2 * br + (q >> 1), c = C0 + 2 * bc + (q & 1);
        lab[q] = sl[q] = -1;
        lv[q] = 0;
        if (r < a.n && c < a.m) {
            const int64_t p = (int64_t)b * P + (int64_t)r * a.m + c;
            lab[q] = a.labels[p];
            sl[q] = a.src_labels[p];
            lv[q] = a.level[p];
        }
    }
    const bool first_only = a.src_first_only && ch == 0;       // (the scalar chunk on its own: a transfer of that chunk alone)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (first_only) x[u].y = x[u].z = x[u].w = 0.0f;
        if (t + RC_T * u < nrs) {
            const float inv = a.src_inv ? 1.0f / sc[u] : 1.0f;
            A[t + RC_T * u] = a.src_inv ? mul4(x[u], inv) : x[u];
            if (ride) A0[t + RC_T * u] = a.src_inv ? x0[u] * inv : x0[u];
        }
    }
    const bool dec = a.posfeat != nullptr;          // chunk 0 of the result = [value | position, size] of the node
    float* dst0 = a.opart[0];
    const int ldd0 = a.opart_w[0];
    auto put = [&](int node, float4 v) {
        if (dec && ch == 0) {
            const float* pf = a.posfeat + 3 * (int64_t)node;
            v.y = pf[0]; v.z = pf[1]; v.w = pf[2];
        }
        *reinterpret_cast<float4*>(dst + (int64_t)node * ldd) = v;
    };
    auto put0 = [&](int node, float v) {            // the rider's result row: (value, position, size) or (gradient, 0, 0, 0)
        float4 o = make_float4(v, 0.0f, 0.0f, 0.0f);
        if (dec) {
            const float* pf = a.posfeat + 3 * (int64_t)node;
            o.y = pf[0]; o.z = pf[1]; o.w = pf[2];
        }
        *reinterpret_cast<float4*>(dst0 + (int64_t)node * ldd0) = o;
    };
    lds_barrier();

    // ---- pixel values, single-pixel nodes, 2 x 2 sums
    float4 v[4];
    float v0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool ok = lab[q] >= 0 && sl[q] >= 0;
        const int row = ok ? ((sl[q] - r0s) & (RC_ROWS - 1)) : 0;
        v[q] = A[row];
        v0[q] = ride ? A0[row] : 0.0f;
        if (!ok) {
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            v0[q] = 0.0f;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (lab[q] >= 0 && lv[q] == 0) {
            put(lab[q], v[q]);
            if (ride) put0(lab[q], v0[q]);
        }
    const int L = lab[0] >= 0 ? lv[0] : 0;          // a node of level >= 1 has its head at a block's first pixel
    float oscale = 1.0f;
    if (L >= 1 && a.mean) oscale = 1.0f / a.npix[lab[0]];          // (requested before the pyramid, used after it)
    // ---- sum pyramid, (top-left + bottom-left) + (top-right + bottom-right) at every level as before: levels 1 .. 4 in registers
    const float4 s1 = add4(add4(v[0], v[1]), add4(v[2], v[3]));
    const float4 s2 = quad_sum4(s1, 0), s3 = quad_sum4(s2, 2), s4 = quad_sum4(s3, 4);
    float r1 = 0.0f, r2 = 0.0f, r3 = 0.0f, r4 = 0.0f;
    if (ride) {
        r1 = (v0[0] + v0[1]) + (v0[2] + v0[3]);
        r2 = quad_sum(r1, 0); r3 = quad_sum(r2, 2); r4 = quad_sum(r3, 4);
    }
    lds_barrier();                                  // every gather done: A / A0 are free
    if ((t & 63) == 0) {
        A[t >> 6] = s4;
        A0[t >> 6] = r4;
    }
    lds_barrier();
    // ---- every node of level >= 1 is written by the thread that owns its head pixel (the block's first pixel, aligned to 2^L)
    if (L >= 1 && ((2 * br) & ((1 << L) - 1)) == 0 && ((2 * bc) & ((1 << L) - 1)) == 0) {
        float4 s;
        float r;
        auto l5 = [&](int j) { return add4(add4(A[4 * j], A[4 * j + 2]), add4(A[4 * j + 1], A[4 * j + 3])); };
        auto l50 = [&](int j) { return (A0[4 * j] + A0[4 * j + 2]) + (A0[4 * j + 1] + A0[4 * j + 3]); };
        switch (L) {
            case 1: s = s1; r = r1; break;
            case 2: s = s2; r = r2; break;
            case 3: s = s3; r = r3; break;
            case 4: s = s4; r = r4; break;
            case 5: s = l5(t >> 8); r = ride ? l50(t >> 8) : 0.0f; break;
            default:
                s = add4(add4(l5(0), l5(2)), add4(l5(1), l5(3)));
                r = ride ? (l50(0) + l50(2)) + (l50(1) + l50(3)) : 0.0f;
                break;
        }
        put(lab[0], mul4(s, oscale));
        if (ride) put0(lab[0], r * oscale);
    }
}

// The same pyramid for image -> mesh pooling of scalar channels (flatten, model/graph_functions.py:391-419: the encoder's input
// frames, the decoder's concat layer, the loss targets): one workgroup per (clip, frame, channel); a pixel's value comes straight
// from the image, so nothing is staged.  img (B, S, n*m, C) with clip stride img_clip_stride; out[(s * N + node) * out_stride +
// out_coff + c].
struct PcArgs {
    const float* img;
    int64_t img_clip_stride;
    int S, C;
    const int32_t* labels;
    const uint8_t* level;
    const float* npix;
    int mean, B, n, m, N, tiles_c, tiles;
    float* out;
    int out_stride, out_coff;
};

__global__ __launch_bounds__(RC_T) void k_pool_clip(PcArgs a) {
    __shared__ float L1[1024], L2[256], L3[64], L4[16], L5[4], L6[1];
    const int t = threadIdx.x;
    const int b = (int)blockIdx.x % a.B;
    const int rest = (int)blockIdx.x / a.B;
    const int tile = rest % a.tiles, sc_ = rest / a.tiles;
    const int R0 = (tile / a.tiles_c) * 64, C0 = (tile % a.tiles_c) * 64;
    const int s = sc_ / a.C, c = sc_ - s * a.C;
    const int P = a.n * a.m;
    const float* img = a.img + (int64_t)b * a.img_clip_stride + (int64_t)s * P * a.C + c;
    float* out = a.out + (int64_t)s * a.N * a.out_stride + a.out_coff + c;
    const int br = (int)compact_bits((unsigned)t), bc = (int)compact_bits((unsigned)t >> 1);
    int lab[4], lv[4];
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = R0 + 2 * br + (q >> 1), cc = C0 + 2 * bc + (q & 1);
        lab[q] = -1;
        lv[q] = 0;
        v[q] = 0.0f;
        if (r < a.n && cc < a.m) {
            const int p = r * a.m + cc;
            lab[q] = a.labels[(int64_t)b * P + p];
            lv[q] = a.level[(int64_t)b * P + p];
            v[q] = img[(int64_t)p * a.C];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (lab[q] < 0) v[q] = 0.0f;
        else if (lv[q] == 0) out[(int64_t)lab[q] * a.out_stride] = v[q];
    }
    const float s1 = (v[0] + v[1]) + (v[2] + v[3]);
    const int L = lab[0] >= 0 ? lv[0] : 0;
    float oscale = 1.0f;
    if (L >= 1 && a.mean) oscale = 1.0f / a.npix[lab[0]];
    L1[t] = s1;
    lds_barrier();
    if (t < 256) L2[t] = (L1[4 * t] + L1[4 * t + 2]) + (L1[4 * t + 1] + L1[4 * t + 3]);
    lds_barrier();
    if (t < 64) L3[t] = (L2[4 * t] + L2[4 * t + 2]) + (L2[4 * t + 1] + L2[4 * t + 3]);
    lds_barrier();
    if (t < 16) L4[t] = (L3[4 * t] + L3[4 * t + 2]) + (L3[4 * t + 1] + L3[4 * t + 3]);
    lds_barrier();
    if (t < 4) L5[t] = (L4[4 * t] + L4[4 * t + 2]) + (L4[4 * t + 1] + L4[4 * t + 3]);
    lds_barrier();
    if (t < 1) L6[0] = (L5[0] + L5[2]) + (L5[1] + L5[3]);
    lds_barrier();
    if (L >= 1 && ((2 * br) & ((1 << L) - 1)) == 0 && ((2 * bc) & ((1 << L) - 1)) == 0) {
        float sum;
        switch (L) {
            case 1: sum = s1; break;
            case 2: sum = L2[t >> 2]; break;
            case 3: sum = L3[t >> 4]; break;
            case 4: sum = L4[t >> 6]; break;
            case 5: sum = L5[t >> 8]; break;
            default: sum = L6[0]; break;
        }
        out[(int64_t)lab[0] * a.out_stride] = sum * oscale;
    }
}

}  // namespace

extern "C" int qt_remesh_clip_rows(void) { return RC_ROWS; }

extern "C" int qt_remesh_clip(const float* const* src_parts, const int* widths, const int* lds, int nparts,
                              const int32_t* src_labels, const float* src_npix, int src_inv, const int32_t* src_cell_off,
                              const int32_t* labels, const uint8_t* level, const float* npix, int mean, int B, int n, int m,
                              float* const* out_parts, const int* out_widths, int nout, const float* posfeat, int src_first_only,
                              void* stream) {
    QT_ARG(src_parts && widths && lds && nparts >= 1 && nparts <= 8 && src_labels && src_cell_off && labels && level && B > 0,
           "bad arguments");
    QT_ARG(n >= 1 && m >= 1, "empty frame");
    QT_ARG(out_parts && out_widths && nout >= 1 && nout <= 8, "bad output parts");
    QT_ARG(!mean || npix, "mean pooling needs npix");
    QT_ARG(!src_inv || src_npix, "src_inv needs src_npix");
    RcArgs a = {};
    int c4 = 0;
    for (int i = 0; i < nparts; ++i) {
        QT_ARG(src_parts[i] && widths[i] > 0 && widths[i] % 4 == 0 && lds[i] % 4 == 0 && lds[i] >= widths[i] &&
               ((uintptr_t)src_parts[i] & 15) == 0, "source parts must be 16-byte aligned with widths / strides that are multiples of 4");
        a.part[i] = src_parts[i];
        a.part_ld[i] = lds[i];
        c4 += widths[i] / 4;
        a.part_end[i] = c4;
    }
    for (int i = nparts; i < 8; ++i) { a.part[i] = nullptr; a.part_ld[i] = 0; a.part_end[i] = 1 << 30; }
    int o4 = 0;
    for (int i = 0; i < nout; ++i) {
        QT_ARG(out_parts[i] && out_widths[i] > 0 && out_widths[i] % 4 == 0 && ((uintptr_t)out_parts[i] & 15) == 0,
               "output parts must be 16-byte aligned with widths that are multiples of 4");
        a.opart[i] = out_parts[i];
        a.opart_w[i] = out_widths[i];
        o4 += out_widths[i] / 4;
        a.opart_end[i] = o4;
    }
    for (int i = nout; i < 8; ++i) { a.opart[i] = nullptr; a.opart_w[i] = 0; a.opart_end[i] = 1 << 30; }
    QT_ARG(o4 == c4, "the output parts must add up to the source width");
    a.src_labels = src_labels;
    a.src_npix = src_npix;
    a.src_off = src_cell_off;
    a.tiles_c = qt_cdiv(m, 64);
    a.tiles = qt_cdiv(n, 64) * a.tiles_c;
    a.labels = labels;
    a.level = level;
    a.npix = npix;
    a.src_inv = src_inv;
    a.mean = mean;
    a.B = B;
    a.n = n;
    a.m = m;
    a.posfeat = posfeat;
    a.src_first_only = src_first_only;
    a.ride = (posfeat || src_first_only) && c4 > 1 && widths[0] == 4 && out_widths[0] == 4 ? 1 : 0;
    QT_ARG((int64_t)B * a.tiles * c4 < ((int64_t)1 << 31), "grid too large");
    hipLaunchKernelGGL(k_remesh_clip, dim3(B * a.tiles * (c4 - a.ride)), dim3(RC_T), 0, (hipStream_t)stream, a);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_pool_clip(const float* img, int S, int64_t img_clip_stride, int C, const int32_t* labels, const uint8_t* level,
                            const float* npix, int mean, int B, int n, int m, int N, float* out, int out_stride, int out_coff,
                            void* stream) {
    QT_ARG(img && labels && level && out && S >= 1 && C >= 1 && B > 0, "bad arguments");
    QT_ARG(n >= 1 && m >= 1, "empty frame");
    QT_ARG(!mean || npix, "mean pooling needs npix");
    QT_ARG(out_stride >= out_coff + C, "output row too short");
    if (N <= 0) return QT_OK;
    const int tiles_c = qt_cdiv(m, 64), tiles = qt_cdiv(n, 64) * tiles_c;
    QT_ARG((int64_t)B * tiles * S * C < ((int64_t)1 << 31), "grid too large");
    PcArgs a = {img, img_clip_stride > 0 ? img_clip_stride : (int64_t)S * n * m * C, S, C, labels, level, npix, mean, B, n, m, N,
                tiles_c, tiles, out, out_stride, out_coff};
    hipLaunchKernelGGL(k_pool_clip, dim3(B * tiles * S * C), dim3(RC_T), 0, (hipStream_t)stream, a);
    QT_LAUNCHED();
    return QT_OK;
}
