// On-device quadtree decomposition (mesh build, stage 1-3) for B clips at once.
// Replaces quadtree_decompose + get_mapping, model/graph_functions.py:145-259, :555-587.
//
// The reference walks the tree depth-first on the host.  Here every base cell is one
// workgroup: a max-pyramid over the (size+1)^2 split windows gives all split decisions
// at once, a pixel's leaf is the first non-splitting ancestor, and the reference's DFS
// label order is recovered as an exclusive prefix sum over leaf heads laid out in
// "reversed Morton" order (children are visited (1,1),(0,1),(1,0),(0,0); base cells in
// reverse row-major order).
#include "qt_common.h"
#include <cstdarg>
#include <cstdio>
#include <math.h>

static thread_local char g_qt_err[512] = "";
void qt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_qt_err, sizeof(g_qt_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* qt_last_error(void) { return g_qt_err; }
extern "C" int qt_abi_version(void) { return 1; }

namespace {

__device__ __forceinline__ unsigned spread_bits(unsigned v) {
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
// child visiting rank: pair value = rowbit + 2*colbit, visited in descending pair order
__device__ __forceinline__ unsigned morton_rc(unsigned r, unsigned c) { return spread_bits(r) | (spread_bits(c) << 1); }

struct Stage1Args {
    const float* src;
    int src_rows, src_cols;
    const float* nodeval;
    int nodeval_stride;
    const int32_t* old_labels;
    int B, n, m, MS, nbi, nbj;
    float thresh;
    int larger, negate;
    const uint8_t* mask;
    const uint8_t* hir;
    int32_t* local_id;
    uint8_t* level;
    int32_t* cnt;
    int32_t* fill;          // optional: fill[0 .. fill_len) = -1 (the old mesh's bwd_src: stage 3 writes only the entries of old nodes
    int fill_len;           // whose head pixel carries their label, every other entry must read "no direct source row")
};

// criterion value and mask / high-interest flags of window element (gr, gc) of clip b, with the reference's clamps
__device__ __forceinline__ void window_elem(const Stage1Args& a, int b, int gr, int gc, int n_pad, int m_pad, int64_t P, float* val,
                                            uint8_t* flag) {
    float v = -INFINITY;
    uint8_t f = 0;
    if (gr < n_pad && gc < m_pad) {
        if (a.src) {
            const int rr = min(gr, a.src_rows - 1), cc = min(gc, a.src_cols - 1);
            v = a.src[((int64_t)b * a.src_rows + rr) * a.src_cols + cc];
        } else {
            const int rr = min(gr, a.n - 1), cc = min(gc, a.m - 1);
            const int lab = a.old_labels[b * P + (int64_t)rr * a.m + cc];
            v = lab >= 0 ? a.nodeval[(int64_t)lab * a.nodeval_stride] : 0.0f;
        }
        if (a.negate) v = -v;
        if (gr < a.n && gc < a.m) {
            const int64_t p = (int64_t)gr * a.m + gc;
            if (a.mask && a.mask[p]) f |= 1;
            if (a.hir && a.hir[p]) f |= 2;
        }
    }
    *val = v;
    *flag = f;
}

__device__ __forceinline__ void fill_neg1(const Stage1Args& a, int nthreads) {
    if (!a.fill) return;
    const int64_t i0 = (int64_t)blockIdx.x * nthreads + threadIdx.x, step = (int64_t)gridDim.x * nthreads;
    for (int64_t i = i0; i < a.fill_len; i += step) a.fill[i] = -1;
}

constexpr int PITCH = 66;

// 1024 threads per base cell: the B * nbase workgroups are few (32 at the bench shape), so the phases below are
// latency chains over the cell's 4096 pixels; with 256 threads the kernel took 39 us.
constexpr int S1T = 1024;
__global__ __launch_bounds__(S1T) void k_quadtree_stage1(Stage1Args a) {
    __shared__ float v[65 * PITCH];
    __shared__ uint8_t fm[65 * PITCH];
    __shared__ float D[1365];
    __shared__ uint8_t Fp[1365];
    __shared__ uint8_t split[1365];
    __shared__ uint8_t lvl[4096];
    __shared__ int flags[4096];
    __shared__ int red[16];

    const int t = threadIdx.x;
    const int MS = a.MS, W1 = MS + 1;
    const int nbase = a.nbi * a.nbj;
    const int b = blockIdx.x / nbase, base = blockIdx.x % nbase;
    const int bi = base / a.nbj, bj = base % a.nbj;
    const int x0 = bi * MS, y0 = bj * MS;
    const int n_pad = a.nbi * MS, m_pad = a.nbj * MS;
    const int64_t P = (int64_t)a.n * a.m;

    // ---- 1. criterion window (MS+1)^2 with the reference's clamps
    fill_neg1(a, S1T);
    for (int idx = t; idx < W1 * W1; idx += S1T) {
        const int r = idx / W1, c = idx % W1;
        window_elem(a, b, x0 + r, y0 + c, n_pad, m_pad, P, &v[r * PITCH + c], &fm[r * PITCH + c]);
    }
    __syncthreads();

    // ---- 2. window-max pyramid; level l has (MS >> l)^2 cells of size 2^l
    int L = 0;
    while ((1 << L) < MS) ++L;
    const float thr = a.negate ? -a.thresh : a.thresh;
    int lbase = 0, prev_base = 0;
    for (int l = 1; l <= L; ++l) {
        const int nc = MS >> l;
        for (int idx = t; idx < nc * nc; idx += S1T) {
            const int i = idx / nc, j = idx % nc;
            float mx;
            uint8_t fo;
            if (l == 1) {
                mx = -INFINITY;
                fo = 0;
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) {
                        const int o = (2 * i + dr) * PITCH + 2 * j + dc;
                        mx = fmaxf(mx, v[o]);
                        fo |= fm[o];
                    }
            } else {
                const int pc = nc * 2;
                const int o = prev_base + (2 * i) * pc + 2 * j;
                mx = fmaxf(fmaxf(D[o], D[o + 1]), fmaxf(D[o + pc], D[o + pc + 1]));
                fo = Fp[o] | Fp[o + 1] | Fp[o + pc] | Fp[o + pc + 1];
            }
            D[lbase + idx] = mx;
            Fp[lbase + idx] = fo;
            const bool s = (a.larger ? (mx > thr) : (mx < thr)) || fo != 0;
            split[lbase + idx] = s ? 1 : 0;
        }
        __syncthreads();
        prev_base = lbase;
        lbase += nc * nc;
    }

    // ---- 3. leaf level of every pixel, head flags in DFS (reversed Morton) order
    const int MS2 = MS * MS;
    for (int idx = t; idx < MS2; idx += S1T) flags[idx] = 0;
    __syncthreads();
    for (int idx = t; idx < MS2; idx += S1T) {
        const int r = idx / MS, c = idx % MS;
        int leaf = 0, off = lbase;
        for (int l = L; l >= 1; --l) {
            const int nc = MS >> l;
            off -= nc * nc;
            if (!split[off + (r >> l) * nc + (c >> l)]) {
                leaf = l;
                break;
            }
        }
        const int s = 1 << leaf;
        const int r0 = r & ~(s - 1), c0 = c & ~(s - 1);
        bool valid = (x0 + r0 < a.n) && (y0 + c0 < a.m);
        if (leaf == 0 && (fm[r * PITCH + c] & 1)) valid = false;
        lvl[idx] = (uint8_t)(leaf | (valid ? 0x80 : 0));
        if (valid && r == r0 && c == c0) flags[MS2 - 1 - (int)morton_rc(r, c)] = 1;
    }
    __syncthreads();

    // ---- 4. exclusive scan over the MS^2 keys
    const int per = (MS2 + S1T - 1) / S1T;
    const int k0 = t * per;
    int sum = 0;
    for (int k = k0; k < min(k0 + per, MS2); ++k) sum += flags[k];
    int total;
    int run = qt_block_excl_scan(sum, red, &total);
    for (int k = k0; k < min(k0 + per, MS2); ++k) {
        const int f = flags[k];
        flags[k] = run;
        run += f;
    }
    __syncthreads();

    // ---- 5. per-pixel local leaf id and level (in-image pixels only)
    for (int idx = t; idx < MS2; idx += S1T) {
        const int r = idx / MS, c = idx % MS;
        const int gr = x0 + r, gc = y0 + c;
        if (gr >= a.n || gc >= a.m) continue;
        const int lv = lvl[idx] & 0x7f;
        const int s = 1 << lv;
        const int r0 = r & ~(s - 1), c0 = c & ~(s - 1);
        const int id = (lvl[idx] & 0x80) ? flags[MS2 - 1 - (int)morton_rc(r0, c0)] : -1;
        const int64_t p = b * P + (int64_t)gr * a.m + gc;
        a.local_id[p] = id;
        a.level[p] = (uint8_t)lv;
    }
    if (t == 0) a.cnt[b * nbase + (nbase - 1 - base)] = total;
}

// The same decomposition with a 64 x 64 base cell split over FOUR workgroups, one per 32 x 32 quadrant: a mesh build has only
// B * nbase base cells (32 at the bench shape: an eighth of the CUs, each walking its 4096 pixels through a dozen barrier-
// separated phases -- 18 us per launch).  A quadrant's workgroup decides the base cell's top-level split itself (max and
// mask / high-interest flags over the whole 65 x 65 window: one load phase, a block reduction) and otherwise sees only its own
// 33 x 33 window: levels 1 .. 5 of the pyramid, the leaf of every pixel, the leaf heads in reversed-Morton (= DFS) order, their
// exclusive scan.  Leaf counts go out per QUADRANT in DFS order -- children are visited (1,1), (0,1), (1,0), (0,0), so quadrant
// (qr, qc) has rank 3 - (qr + 2 qc) -- and stage 3's scan over the counts adds the offsets; a base cell that does not split is
// one leaf, counted by the rank-0 workgroup and recognised in stage 3 by its level (6).
constexpr int QS = 32, QW = QS + 1, QPITCH = 34, QT = 1024;
__global__ __launch_bounds__(QT) void k_quadtree_stage1q(Stage1Args a) {
    __shared__ float v[QW * QPITCH];
    __shared__ uint8_t fm[QW * QPITCH];
    __shared__ float D[341];
    __shared__ uint8_t Fp[341];
    __shared__ uint8_t split[341];
    __shared__ int flags[QS * QS];
    __shared__ int red[16];
    __shared__ float redf[16];
    __shared__ int redo[16];

    const int t = threadIdx.x;
    const int MS = 64;
    const int nbase = a.nbi * a.nbj;
    const int quad = blockIdx.x & 3, cellid = blockIdx.x >> 2;
    const int b = cellid / nbase, base = cellid % nbase;
    const int bi = base / a.nbj, bj = base % a.nbj;
    const int qr = quad >> 1, qc = quad & 1;
    const int X0 = bi * MS, Y0 = bj * MS;                 // base cell origin
    const int x0 = X0 + qr * QS, y0 = Y0 + qc * QS;       // quadrant origin
    const int n_pad = a.nbi * MS, m_pad = a.nbj * MS;
    const int64_t P = (int64_t)a.n * a.m;
    const float thr = a.negate ? -a.thresh : a.thresh;
    fill_neg1(a, QT);

    // ---- 1. the quadrant's own (QS+1)^2 window into LDS; max / flags over the base cell's whole (MS+1)^2 window in registers
    for (int idx = t; idx < QW * QW; idx += QT) {
        const int r = idx / QW, c = idx % QW;
        window_elem(a, b, x0 + r, y0 + c, n_pad, m_pad, P, &v[r * QPITCH + c], &fm[r * QPITCH + c]);
    }
    float wmax = -INFINITY;
    int wflag = 0;
    for (int idx = t; idx < (MS + 1) * (MS + 1); idx += QT) {
        const int r = idx / (MS + 1), c = idx % (MS + 1);
        float val;
        uint8_t f;
        window_elem(a, b, X0 + r, Y0 + c, n_pad, m_pad, P, &val, &f);
        wmax = fmaxf(wmax, val);
        wflag |= f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        wmax = fmaxf(wmax, __shfl_xor(wmax, d, 64));
        wflag |= __shfl_xor(wflag, d, 64);
    }
    if ((t & 63) == 0) {
        redf[t >> 6] = wmax;
        redo[t >> 6] = wflag;
    }
    __syncthreads();
    float tmax = -INFINITY;
    int tflag = 0;
#pragma unroll
    for (int w = 0; w < QT / 64; ++w) {
        tmax = fmaxf(tmax, redf[w]);
        tflag |= redo[w];
    }
    const bool top_split = (a.larger ? (tmax > thr) : (tmax < thr)) || tflag != 0;
    const int rank = 3 - (qr + 2 * qc);
    const int cnt_slot = (b * nbase + (nbase - 1 - base)) * 4 + rank;
    if (!top_split) {
        // the base cell is one leaf (level 6): every in-image pixel gets local id 0 relative to the cell's first count slot
        const int r = t >> 5, c = t & 31;
        const int gr = x0 + r, gc = y0 + c;
        if (gr < a.n && gc < a.m) {
            const int64_t p = b * P + (int64_t)gr * a.m + gc;
            a.local_id[p] = 0;
            a.level[p] = 6;
        }
        if (t == 0) a.cnt[cnt_slot] = (rank == 0 && X0 < a.n && Y0 < a.m) ? 1 : 0;
        return;
    }

    // ---- 2. window-max pyramid of the quadrant; level l has (QS >> l)^2 cells of size 2^l, l = 1 .. 5
    int lbase = 0, prev_base = 0;
    for (int l = 1; l <= 5; ++l) {
        const int nc = QS >> l;
        for (int idx = t; idx < nc * nc; idx += QT) {
            const int i = idx / nc, j = idx % nc;
            float mx;
            uint8_t fo;
            if (l == 1) {
                mx = -INFINITY;
                fo = 0;
#pragma unroll
                for (int dr = 0; dr < 3; ++dr)
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) {
                        const int o = (2 * i + dr) * QPITCH + 2 * j + dc;
                        mx = fmaxf(mx, v[o]);
                        fo |= fm[o];
                    }
            } else {
                // (a level-l window is the union of its four children's windows: they overlap in the rows / columns the
                // children share and reach the same extra row and column)
                const int pc = nc * 2;
                const int o = prev_base + (2 * i) * pc + 2 * j;
                mx = fmaxf(fmaxf(D[o], D[o + 1]), fmaxf(D[o + pc], D[o + pc + 1]));
                fo = Fp[o] | Fp[o + 1] | Fp[o + pc] | Fp[o + pc + 1];
            }
            D[lbase + idx] = mx;
            Fp[lbase + idx] = fo;
            split[lbase + idx] = ((a.larger ? (mx > thr) : (mx < thr)) || fo != 0) ? 1 : 0;
        }
        __syncthreads();
        prev_base = lbase;
        lbase += nc * nc;
    }

    // ---- 3. leaf level of this thread's pixel, head flags in DFS (reversed Morton) order
    const int r = t >> 5, c = t & 31;
    flags[t] = 0;
    __syncthreads();
    int leaf = 0, off = lbase;
    for (int l = 5; l >= 1; --l) {
        const int nc = QS >> l;
        off -= nc * nc;
        if (!split[off + (r >> l) * nc + (c >> l)]) {
            leaf = l;
            break;
        }
    }
    const int sz = 1 << leaf;
    const int r0 = r & ~(sz - 1), c0 = c & ~(sz - 1);
    bool valid = (x0 + r0 < a.n) && (y0 + c0 < a.m);
    if (leaf == 0 && (fm[r * QPITCH + c] & 1)) valid = false;
    if (valid && r == r0 && c == c0) flags[QS * QS - 1 - (int)morton_rc(r, c)] = 1;
    __syncthreads();

    // ---- 4. exclusive scan over the QS^2 keys (one per thread)
    const int f = flags[t];
    int total;
    const int ex = qt_block_excl_scan(f, red, &total);
    __syncthreads();
    flags[t] = ex;
    __syncthreads();

    // ---- 5. local leaf id (relative to the quadrant's count slot) and level of this thread's pixel
    const int gr = x0 + r, gc = y0 + c;
    if (gr < a.n && gc < a.m) {
        const int id = valid ? flags[QS * QS - 1 - (int)morton_rc(r0, c0)] : -1;
        const int64_t p = b * P + (int64_t)gr * a.m + gc;
        a.local_id[p] = id;
        a.level[p] = (uint8_t)leaf;
    }
    if (t == 0) a.cnt[cnt_slot] = total;
}

// positional encoding + size feature of a node (add_positional_encoding and the size channel, model/graph_functions.py:366-389,
// :599-607): centroid / image extent, pixel count / size_norm
__device__ __forceinline__ void node_features(int4 cl, int i, int n, int m, float size_norm, float* __restrict__ feat,
                                              float* __restrict__ npix) {
    const int hh = min(cl.x + cl.z, n) - cl.x, ww = min(cl.y + cl.z, m) - cl.y;
    const float np_ = (float)(hh * ww);
    feat[3 * i + 0] = ((float)cl.y + 0.5f * (float)(ww - 1)) / (float)m;
    feat[3 * i + 1] = ((float)cl.x + 0.5f * (float)(hh - 1)) / (float)n;
    feat[3 * i + 2] = np_ / size_norm;
    npix[i] = np_;
}

__global__ void k_quadtree_stage3(const int32_t* __restrict__ local_id, const int32_t* __restrict__ offs_in, int raw, int quads,
                                  int B, int n, int m, int MS, int nbj, int nbase,
                                  int32_t* __restrict__ labels, const uint8_t* __restrict__ level,
                                  int32_t* __restrict__ cell, int32_t* __restrict__ node_off, float size_norm,
                                  float* __restrict__ feat, float* __restrict__ npix,
                                  const int32_t* __restrict__ old_labels, const uint8_t* __restrict__ old_level,
                                  int32_t* __restrict__ fwd_src, int32_t* __restrict__ bwd_src, int32_t* __restrict__ cell_off) {
    // raw != 0: offs_in holds the per-cell leaf COUNTS (stage 1's output) and every workgroup scans the B * nbase <= 1024 of
    // them itself in LDS -- the separate scan launch between the two stages is gone (static capacities: nobody on the host
    // needs the total)
    __shared__ int soffs[1025];
    __shared__ int red[8];
    const int32_t* offs = offs_in;
    // quads != 0: the counts come per 32 x 32 quadrant of a 64 x 64 base cell (k_quadtree_stage1q), four slots per base cell in
    // DFS order; a pixel of level 6 (its base cell is one leaf) counts from the cell's first slot
    const int per = quads ? 4 : 1;
    if (raw) {
        const int len = B * nbase * per;
        int carry = 0;
        for (int c0 = 0; c0 < len; c0 += 256) {
            const int i = c0 + (int)threadIdx.x;
            const int vv = i < len ? offs_in[i] : 0;
            int total;
            const int ex = qt_block_excl_scan_256(vv, red, &total);
            if (i < len) soffs[i] = carry + ex;
            carry += total;
        }
        if (threadIdx.x == 0) soffs[len] = carry;
        __syncthreads();
        offs = soffs;
    }
    const int64_t P = (int64_t)n * m;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx <= B) node_off[idx] = offs[idx * nbase * per];
    // first node of every base cell, in label order (slot b * nbase + (nbase - 1 - base)): a base cell's nodes are one contiguous
    // label range, which is what lets the clip-resident transfer (remeshclip.hip) stage one 64 x 64 tile's source rows
    if (cell_off && idx <= (int64_t)B * nbase) cell_off[idx] = offs[idx * per];
    if (idx >= B * P) return;
    const int b = (int)(idx / P);
    const int p = (int)(idx % P);
    const int r = p / m, c = p % m;
    const int base = (r / MS) * nbj + (c / MS);
    const int id = local_id[idx];
    int lab = -1;
    if (id >= 0) {
        int slot = b * nbase + (nbase - 1 - base);
        if (quads) {
            const int rank = level[idx] >= 6 ? 0 : 3 - (((r % MS) >> 5) + 2 * ((c % MS) >> 5));
            slot = slot * 4 + rank;
        }
        lab = id + offs[slot];
        const int s = 1 << level[idx];
        if ((r & (s - 1)) == 0 && (c & (s - 1)) == 0) {
            int4 cl = make_int4(r, c, s, b);
            reinterpret_cast<int4*>(cell)[lab] = cl;
            if (feat) node_features(cl, lab, n, m, size_norm, feat, npix);
            // state transfer old mesh -> this mesh: a single-pixel node takes the row of the old node under its pixel
            // (qt_remesh reads this index instead of walking cell -> pixel -> old label); larger nodes: -1 = general path
            if (fwd_src) fwd_src[lab] = s == 1 ? old_labels[idx] : -1;
        }
    }
    if (bwd_src) {
        // ... and the transposed transfer (the gradient back onto the old mesh): for the old node whose head pixel this is
        const int ol = old_labels[idx];
        if (ol >= 0) {
            const int so = 1 << old_level[idx];
            if ((r & (so - 1)) == 0 && (c & (so - 1)) == 0) bwd_src[ol] = so == 1 ? lab : -1;
        }
    }
    labels[idx] = lab;
}

// ---- three-kernel exclusive scan (2048 items per 256-thread workgroup)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = 256 * SCAN_ITEMS;

__global__ __launch_bounds__(256) void k_scan_sums(const int32_t* __restrict__ in, int64_t len, int32_t* __restrict__ sums) {
    __shared__ int red[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < len) s += in[base + i];
    int total;
    qt_block_excl_scan_256(s, red, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: in-place exclusive scan of sums[0..nblk), sums[nblk] = grand total
__global__ __launch_bounds__(256) void k_scan_top(int32_t* __restrict__ sums, int nblk) {
    __shared__ int red[8];
    int carry = 0;
    for (int c0 = 0; c0 < nblk; c0 += 256) {
        const int i = c0 + threadIdx.x;
        const int vv = i < nblk ? sums[i] : 0;
        int total;
        const int ex = qt_block_excl_scan_256(vv, red, &total);
        if (i < nblk) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[nblk] = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const int32_t* __restrict__ in, int64_t len,
                                                    const int32_t* __restrict__ sums, int32_t* __restrict__ out) {
    __shared__ int red[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    int vals[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        vals[i] = (base + i < len) ? in[base + i] : 0;
        s += vals[i];
    }
    int total;
    int run = sums[blockIdx.x] + qt_block_excl_scan_256(s, red, &total);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < len) out[base + i] = run;
        run += vals[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[len] = sums[gridDim.x];
}

// one workgroup: out[0] = 0, out[i+1] = in[0] + .. + in[i]
__global__ __launch_bounds__(256) void k_scan_small(const int32_t* __restrict__ in, int len, int32_t* __restrict__ out) {
    __shared__ int red[8];
    int carry = 0;
    for (int c0 = 0; c0 < len; c0 += SCAN_CHUNK) {
        const int base = c0 + threadIdx.x * SCAN_ITEMS;
        int vals[SCAN_ITEMS];
        int sum = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            vals[i] = (base + i < len) ? in[base + i] : 0;
            sum += vals[i];
        }
        int total;
        int run = carry + qt_block_excl_scan_256(sum, red, &total);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            if (base + i < len) out[base + i] = run;
            run += vals[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0) out[len] = carry;
}

__global__ void k_node_features(const int32_t* __restrict__ cell, int Ncap, const int32_t* __restrict__ n_dev, int n,
                                int m, float size_norm, float* __restrict__ feat, float* __restrict__ npix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qt_rows(n_dev, Ncap)) return;
    node_features(reinterpret_cast<const int4*>(cell)[i], i, n, m, size_norm, feat, npix);
}

}  // namespace

extern "C" int qt_quadtree_stage1(const float* src, int src_rows, int src_cols, const float* nodeval,
                                  int nodeval_stride, const int32_t* old_labels, int B, int n, int m, int max_size, float thresh,
                                  int condition, const uint8_t* mask, const uint8_t* hir, int32_t* local_id,
                                  uint8_t* level, int32_t* cnt, int quads, int32_t* fill_neg1_ptr, int fill_len, void* stream) {
    QT_ARG(B > 0 && n > 0 && m > 0, "empty image batch");
    QT_ARG(max_size >= 2 && max_size <= 64 && (max_size & (max_size - 1)) == 0, "max_size must be a power of two in [2, 64]");
    QT_ARG(condition >= 0 && condition <= 3, "unknown condition");
    QT_ARG((src != nullptr) != (nodeval != nullptr && old_labels != nullptr), "give either src or nodeval+old_labels");
    QT_ARG(local_id && level && cnt, "null output");
    QT_ARG(!quads || max_size == 64, "quads: one workgroup per 32 x 32 quadrant needs max_size == 64");
    QT_ARG(fill_len >= 0 && (fill_len == 0 || fill_neg1_ptr), "fill_len without a buffer");
    Stage1Args a;
    a.src = src;
    a.src_rows = src_rows;
    a.src_cols = src_cols;
    a.nodeval = nodeval;
    a.nodeval_stride = nodeval_stride > 0 ? nodeval_stride : 1;
    a.old_labels = old_labels;
    a.B = B;
    a.n = n;
    a.m = m;
    a.MS = max_size;
    a.nbi = qt_cdiv(n, max_size);
    a.nbj = qt_cdiv(m, max_size);
    QT_ARG(a.nbi <= a.nbj, "padded rows exceed padded columns (the reference raises IndexError here)");
    if (src) QT_ARG(src_rows >= 1 && src_cols >= 1, "bad src shape");
    a.thresh = thresh;
    a.larger = (condition == QT_COND_MAX_LARGER || condition == QT_COND_MIN_SMALLER);
    a.negate = (condition >= 2);
    a.mask = mask;
    a.hir = hir;
    a.local_id = local_id;
    a.level = level;
    a.cnt = cnt;
    a.fill = fill_len > 0 ? fill_neg1_ptr : nullptr;
    a.fill_len = fill_len;
    if (quads)
        hipLaunchKernelGGL(k_quadtree_stage1q, dim3(4 * B * a.nbi * a.nbj), dim3(QT), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_quadtree_stage1, dim3(B * a.nbi * a.nbj), dim3(S1T), 0, (hipStream_t)stream, a);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_quadtree_stage3(const int32_t* local_id, const int32_t* cnt_offsets, int B, int n, int m,
                                  int max_size, int32_t* labels, const uint8_t* level, int32_t* cell,
                                  int32_t* node_off, float size_norm, float* feat, float* npix, int raw_counts, int quads,
                                  const int32_t* old_labels, const uint8_t* old_level, int32_t* fwd_src, int32_t* bwd_src,
                                  int32_t* cell_off, void* stream) {
    QT_ARG(local_id && cnt_offsets && labels && level && cell && node_off, "null pointer");
    QT_ARG((!fwd_src && !bwd_src) || (old_labels && old_level), "fwd_src / bwd_src need the old mesh's labels and levels");
    QT_ARG((feat == nullptr) == (npix == nullptr), "give both feat and npix or neither");
    const int nbi = qt_cdiv(n, max_size), nbj = qt_cdiv(m, max_size);
    QT_ARG(!quads || max_size == 64, "quads needs max_size == 64");
    QT_ARG(!raw_counts || (int64_t)B * nbi * nbj * (quads ? 4 : 1) <= 1024, "raw_counts: at most 1024 counts (scan them with qt_scan_i32 instead)");
    const int64_t total = (int64_t)B * n * m;
    hipLaunchKernelGGL(k_quadtree_stage3, dim3(qt_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, local_id,
                       cnt_offsets, raw_counts, quads, B, n, m, max_size, nbj, nbi * nbj, labels, level, cell, node_off, size_norm, feat, npix,
                       old_labels, old_level, fwd_src, bwd_src, cell_off);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_scan_i32(const int32_t* in, int32_t* out, int64_t len, int32_t* tmp, void* stream) {
    QT_ARG(in && out && tmp && len > 0, "bad scan arguments");
    const int nblk = qt_cdiv(len, SCAN_CHUNK);
    hipStream_t s = (hipStream_t)stream;
    if (len <= 16 * SCAN_CHUNK) {            // the per-base-cell leaf counts: one workgroup, one launch
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, s, in, (int)len, out);
        QT_LAUNCHED();
        return QT_OK;
    }
    hipLaunchKernelGGL(k_scan_sums, dim3(nblk), dim3(256), 0, s, in, len, tmp);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, s, tmp, nblk);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblk), dim3(256), 0, s, in, len, tmp, out);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_scan_top(int32_t* sums, int nblk, void* stream) {
    QT_ARG(sums && nblk >= 0, "bad arguments");
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, nblk);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_node_features(const int32_t* cell, int N, const int32_t* n_dev, int n, int m, float size_norm,
                                float* feat, float* npix, void* stream) {
    QT_ARG(cell && feat && npix, "null pointer");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_node_features, dim3(qt_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, cell, N, n_dev,
                       n, m, size_norm, feat, npix);
    QT_LAUNCHED();
    return QT_OK;
}
