// Mesh adjacency as CSR, emitted on the device with no sort.
// Replaces get_adj + dist (model/graph_functions.py:261-363) and the per-call ChebConv
// normalisation of PyG (restated in oracle/qt_oracle.py:cheb_norm).
//
// A quadtree leaf is a square, so each neighbour touches exactly one of its four sides
// in one contiguous run: one thread per (node, side) walks the pixels just outside that
// side and counts label changes.  A prefix sum over the 4N counts is the CSR row
// pointer, a second walk fills columns and centroid distances.  Self pairs (n, n),
// which the reference emits for multi-pixel cells, carry weight 0 in every supported
// convolution and are not stored (the Python mirror re-creates them for edge_index).
#include "qt_common.h"

namespace {

struct SideWalk {
    int b, r, c, dr, dc, len;
};

__device__ __forceinline__ SideWalk side_walk(int4 cl, int side, int n, int m) {
    SideWalk w;
    w.b = cl.w;
    const int hh = min(cl.x + cl.z, n) - cl.x, ww = min(cl.y + cl.z, m) - cl.y;
    w.len = 0;
    w.r = w.c = w.dr = w.dc = 0;
    if (side == 0) {  // row above
        if (cl.x > 0) { w.r = cl.x - 1; w.c = cl.y; w.dc = 1; w.len = ww; }
    } else if (side == 1) {  // row below
        if (cl.x + cl.z < n) { w.r = cl.x + cl.z; w.c = cl.y; w.dc = 1; w.len = ww; }
    } else if (side == 2) {  // column to the left
        if (cl.y > 0) { w.r = cl.x; w.c = cl.y - 1; w.dr = 1; w.len = hh; }
    } else {  // column to the right
        if (cl.y + cl.z < m) { w.r = cl.x; w.c = cl.y + cl.z; w.dr = 1; w.len = hh; }
    }
    return w;
}

__device__ __forceinline__ void centroid(int4 cl, int n, int m, float res, float* xx, float* yy) {
    const int hh = min(cl.x + cl.z, n) - cl.x, ww = min(cl.y + cl.z, m) - cl.y;
    *xx = ((float)cl.y + 0.5f * (float)(ww - 1)) * res;
    *yy = ((float)cl.x + 0.5f * (float)(hh - 1)) * res;
}

// One (node, side) per thread, 1024 per workgroup.  The count pass also leaves the per-workgroup totals (`sums`), the
// fill pass turns them (after qt_scan_top) plus a workgroup scan of its own counts into edge offsets: no scan kernels
// over the 4N counts, and the degree normalisation 1/sqrt(sum of the row's weights) falls out of the fill walk.
constexpr int ET = 1024;

__global__ __launch_bounds__(ET) void k_edges_count(const int32_t* __restrict__ labels, const int32_t* __restrict__ cell,
                                                    int Ncap, const int32_t* __restrict__ n_dev, int n, int m,
                                                    int32_t* __restrict__ cnt4, int32_t* __restrict__ sums,
                                                    int32_t* __restrict__ tail_cnt, int B, int32_t* __restrict__ zero_buf,
                                                    int zero_len) {
    __shared__ int red[16];
    const int idx = blockIdx.x * ET + threadIdx.x;
    for (int z = idx; z < zero_len; z += (int)gridDim.x * ET) zero_buf[z] = 0;      // (tile counters + sync words of multi-tile frames)
    if (tail_cnt) {                      // the per-clip counters k_edges_nrm adds to (two launches later): tail edges, rows with a tail
        for (int c = idx; c < B; c += (int)gridDim.x * ET) {      // (grid-stride: any number of clips)
            tail_cnt[QT_TAIL_CNT_STRIDE * c] = 0;
            tail_cnt[QT_TAIL_CNT_STRIDE * c + 1] = 0;
        }
    }
    int cnt = 0;
    if (idx < 4 * qt_rows(n_dev, Ncap)) {
        const int4 cl = reinterpret_cast<const int4*>(cell)[idx >> 2];
        const SideWalk w = side_walk(cl, idx & 3, n, m);
        const int32_t* L = labels + (int64_t)w.b * n * m;
        int prev = -1;
        for (int k = 0; k < w.len; ++k) {
            const int lab = L[(int64_t)(w.r + k * w.dr) * m + (w.c + k * w.dc)];
            if (lab >= 0 && lab != prev) ++cnt;
            prev = lab;
        }
    }
    if (idx < 4 * Ncap) cnt4[idx] = cnt;      // capacity rows beyond N: zero, the offsets stay flat
    int total;
    qt_block_excl_scan(cnt, red, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(ET) void k_edges_fill(const int32_t* __restrict__ labels, const int32_t* __restrict__ cell,
                                                   const int32_t* __restrict__ cnt4, const int32_t* __restrict__ sums,
                                                   int Ncap, const int32_t* __restrict__ n_dev, int n, int m, float res,
                                                   int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
                                                   float* __restrict__ wgt, float* __restrict__ dis) {
    __shared__ int red[16];
    const int idx = blockIdx.x * ET + threadIdx.x;
    const int N = qt_rows(n_dev, Ncap);
    const int mine = idx < 4 * Ncap ? cnt4[idx] : 0;
    // offset of this workgroup = the totals of the workgroups before it (`sums` as qt_edges_count left it: a few hundred
    // entries, one strided pass -- the separate single-workgroup scan launch is gone)
    int before = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += ET) before += sums[j];
    int offset;
    qt_block_excl_scan(before, red, &offset);
    int total;
    int e = offset + qt_block_excl_scan(mine, red, &total);
    const int node = idx >> 2;
    if ((idx & 3) == 0 && node <= Ncap) rowptr[node] = e;       // rows N .. Ncap all point at the end (empty rows)
    float wsum = 0.0f;
    if (idx < 4 * N) {
        const int4 cl = reinterpret_cast<const int4*>(cell)[node];
        const SideWalk w = side_walk(cl, idx & 3, n, m);
        const int32_t* L = labels + (int64_t)w.b * n * m;
        float xx, yy;
        centroid(cl, n, m, res, &xx, &yy);
        int prev = -1;
        for (int k = 0; k < w.len; ++k) {
            const int lab = L[(int64_t)(w.r + k * w.dr) * m + (w.c + k * w.dc)];
            if (lab >= 0 && lab != prev) {
                float x2, y2;
                centroid(reinterpret_cast<const int4*>(cell)[lab], n, m, res, &x2, &y2);
                const float d = sqrtf((yy - y2) * (yy - y2) + (xx - x2) * (xx - x2));
                col[e] = lab;
                wgt[e] = d;
                wsum += d;
                ++e;
            }
            prev = lab;
        }
    }
    // the four sides of a node are four adjacent lanes: fixed-order sum (s0 + s1) + (s2 + s3)
    wsum += __shfl_xor(wsum, 1, 64);
    wsum += __shfl_xor(wsum, 2, 64);
    if ((idx & 3) == 0 && idx < 4 * N) dis[node] = wsum > 0.0f ? 1.0f / sqrtf(wsum) : 0.0f;
}

__global__ void k_edges_nrm(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
                            const float* __restrict__ dis, int Ncap, const int32_t* __restrict__ n_dev,
                            float* __restrict__ nrm, int4* __restrict__ ell, const int32_t* __restrict__ cell,
                            const int32_t* __restrict__ node_off, int32_t* __restrict__ tail_cnt, int2* __restrict__ tail_pool,
                            uint32_t* __restrict__ tail_info, int4* __restrict__ tail_rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qt_rows(n_dev, Ncap)) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    // The edges beyond the fourth again, per CLIP, as {column relative to the clip's first node, weight}: the clip-resident
    // recurrence kernel (chebclip.hip) copies a clip's pool into LDS once per launch.  A row reserves its run with one atomic add
    // on the clip's counter (zeroed by k_edges_count) -- issued here, before the edge loop, so that its round trip overlaps the
    // loop's own loads -- and the loop writes the run as it goes.  The order of the runs varies from build to build, the
    // contents of a run (CSR order) do not.  A clip with more than QT_TAIL_CAP tail edges keeps the rows that did not fit on the
    // CSR walk (base 0xffff).  The counters sit QT_TAIL_CNT_STRIDE ints (one 128-byte line) apart: side by side in one line, the
    // ~3500 atomic adds of a 32-clip mesh serialised on that line and the launch took 32 us instead of 8.
    // A row with a tail also takes one slot of its clip's list of RECORDS (second counter of the clip's line): the row's first
    // four edges as the kernel wants them (columns relative to the clip, packed), its pool descriptor and its own number in the
    // clip -- chebclip.hip hands record j to thread j, so the tail loops run in a few full waves instead of one or two lanes
    // of every wave.  Order of the records: as the atomics arrive; their contents do not depend on it.
    uint32_t info = 0;
    int2* run = nullptr;
    int4* rec = nullptr;
    int r0 = 0;
    if (tail_info) {
        const int cnt = min(e1 - e0 - 4, 0xffff);
        if (cnt > 0) {
            const int clip = cell[4 * (int64_t)i + 3];
            r0 = node_off[clip];
            const int base = atomicAdd(&tail_cnt[QT_TAIL_CNT_STRIDE * clip], cnt);
            if (base + cnt <= QT_TAIL_CAP) {
                run = tail_pool + (int64_t)clip * QT_TAIL_CAP + base;
                info = (uint32_t)base | ((uint32_t)cnt << 16);
            } else {
                info = 0xffffu | ((uint32_t)cnt << 16);
            }
            if (tail_rec) {
                const int slot = atomicAdd(&tail_cnt[QT_TAIL_CNT_STRIDE * clip + 1], 1);
                if (slot < QT_TAIL_REC_CAP) rec = tail_rec + 2 * ((int64_t)clip * QT_TAIL_REC_CAP + slot);
            }
        }
    }
    const float di = dis[i];
    int c4[4] = {i, i, i, i};              // (an unused slot re-reads the row itself with weight 0)
    float w4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int e = e0; e < e1; ++e) {
        const int cj = col[e];
        const float v = -(di * w[e] * dis[cj]);
        nrm[e] = v;
        if (e - e0 < 4) {
            c4[e - e0] = cj;
            w4[e - e0] = v;
        } else if (run && e - e0 - 4 < 0xffff) {
            run[e - e0 - 4] = make_int2(cj - r0, __float_as_int(v));
        }
    }
    if (ell) {
        // the first four edges of every row again, as two 16-byte vectors per row: k_spmm reads them without the row pointer
        // (most quadtree rows have exactly four neighbours); a negative (complemented) last column flags more edges in the CSR
        if (e1 - e0 > 4) c4[3] = ~c4[3];
        ell[2 * (int64_t)i] = make_int4(c4[0], c4[1], c4[2], c4[3]);
        ell[2 * (int64_t)i + 1] = make_int4(__float_as_int(w4[0]), __float_as_int(w4[1]), __float_as_int(w4[2]), __float_as_int(w4[3]));
    }
    if (tail_info) tail_info[i] = info;
    if (rec) {
        const int c3 = c4[3] < 0 ? ~c4[3] : c4[3];
        const unsigned M = QT_TAIL_REC_CAP - 1;
        rec[0] = make_int4((int)((((unsigned)(c4[0] - r0) & M) << 4) | (((unsigned)(c4[1] - r0) & M) << 20)),
                           (int)((((unsigned)(c4[2] - r0) & M) << 4) | (((unsigned)(c3 - r0) & M) << 20)),
                           __float_as_int(w4[0]), __float_as_int(w4[1]));
        rec[1] = make_int4(__float_as_int(w4[2]), __float_as_int(w4[3]), (int)info, i - r0);
    }
}

// qt_edges_norm for frames of several 64 x 64 base cells ("tiles"): nrm / ell as k_edges_nrm, and per TILE the structures the
// tile-resident recurrence kernel (chebclip.hip, TILE = true) runs from.  A tile's nodes are one contiguous label range
// [t0, t0 + nr) (qt_quadtree_stage3's cell_off).  A row is
//   * a BOUNDARY row when at least one of its edges leaves the tile: it gets a boundary record (first four edges, local rows or
//     halo slots) + a run in the boundary pool; every edge that leaves the tile takes one HALO slot of the tile (the global row
//     of the neighbour goes to tile_halo) -- at most 256 of each per tile: a boundary row owns at least one of the tile's
//     <= 256 pixel adjacencies across its border, and so does every (row, outside neighbour) pair;
//   * else, with more than four edges, an INTERIOR record + a run in the interior pool (k_edges_nrm's scheme per tile).
// Slot / run order: as the atomics arrive; contents do not depend on it.
__global__ void k_edges_nrm_tile(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
                                 const float* __restrict__ dis, int Ncap, const int32_t* __restrict__ n_dev,
                                 float* __restrict__ nrm, int4* __restrict__ ell, const int32_t* __restrict__ cell,
                                 const int32_t* __restrict__ tile_off, int T, int nbj, int32_t* __restrict__ tile_cnt,
                                 int2* __restrict__ tile_pool, int4* __restrict__ tile_rec, int4* __restrict__ tile_brec,
                                 int2* __restrict__ tile_bpool, int32_t* __restrict__ tile_halo,
                                 int32_t* __restrict__ brec_addr, int32_t* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= qt_rows(n_dev, Ncap)) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    const int4 cl = reinterpret_cast<const int4*>(cell)[i];
    const int base = (cl.x >> 6) * nbj + (cl.y >> 6);
    const int ts = cl.w * T + (T - 1 - base);                  // tile slot in label order (cell_off's slots)
    const int t0 = tile_off[ts], nr = tile_off[ts + 1] - t0;
    int32_t* cnts = tile_cnt + (int64_t)QT_TILE_CNT_STRIDE * ts;
    int nrem = 0;
    for (int e = e0; e < e1; ++e) {
        const unsigned lj = (unsigned)(col[e] - t0);
        nrem += lj >= (unsigned)nr;
    }
    const int cnt = min(max(e1 - e0 - 4, 0), 0xffff);
    uint32_t info = 0;
    int2* run = nullptr;
    int4* rec = nullptr;
    int hbase = 0;
    if (nrem > 0) {                                            // boundary row
        hbase = atomicAdd(&cnts[2], nrem);
        const int slot = atomicAdd(&cnts[3], 1);
        int pbase = 0;
        if (cnt > 0) pbase = atomicAdd(&cnts[4], cnt);
        if (hbase + nrem <= QT_TILE_HALO_CAP && slot < QT_TILE_HALO_CAP && pbase + cnt <= QT_TILE_BPOOL_CAP) {
            rec = tile_brec + 2 * ((int64_t)ts * QT_TILE_HALO_CAP + slot);
            run = tile_bpool + (int64_t)ts * QT_TILE_BPOOL_CAP + pbase;
            info = (uint32_t)pbase | ((uint32_t)cnt << 16);
            brec_addr[i] = ts * QT_TILE_HALO_CAP + slot;        // where the other tiles find this row's published values
        } else {
            // (cannot happen on a quadtree mesh -- but this entry accepts any CSR: the row gets no record, the other tiles find a
            // sentinel instead of an address and do not wait for it, and the caller's error word says so)
            cnts[5] = 1;
            brec_addr[i] = -1;
            if (err) atomicOr(err, 2);
            // what this row took from the counters stays addressable: a NULL record (info = ~0: the kernel computes and stores
            // nothing for it) in its slot, and its own row number in its halo slots (a valid row whose address is the sentinel)
            if (slot < QT_TILE_HALO_CAP) {
                int4* nul = tile_brec + 2 * ((int64_t)ts * QT_TILE_HALO_CAP + slot);
                nul[0] = make_int4(0, 0, 0, 0);
                nul[1] = make_int4(0, 0, -1, 0);
            }
            for (int h = hbase; h < min(hbase + nrem, QT_TILE_HALO_CAP); ++h) tile_halo[(int64_t)ts * QT_TILE_HALO_CAP + h] = i;
            nrem = 0;
        }
    } else if (cnt > 0) {                                      // interior row with a tail
        const int pbase = atomicAdd(&cnts[0], cnt);
        if (pbase + cnt <= QT_TILE_POOL_CAP) {
            run = tile_pool + (int64_t)ts * QT_TILE_POOL_CAP + pbase;
            info = (uint32_t)pbase | ((uint32_t)cnt << 16);
        } else {
            info = 0xffffu | ((uint32_t)cnt << 16);            // pool full: the row walks the CSR arrays
        }
        const int slot = atomicAdd(&cnts[1], 1);
        if (slot < QT_TILE_REC_CAP) rec = tile_rec + 2 * ((int64_t)ts * QT_TILE_REC_CAP + slot);
        else if (err) atomicOr(err, 2);                        // (more rows with a tail than rows: not a tile of <= 4096 nodes)
    }
    const float di = dis[i];
    int c4[4] = {i, i, i, i};              // (an unused slot re-reads the row itself with weight 0)
    float w4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned l4[4];                        // the first four columns as the kernel's 16-bit fields
    l4[0] = l4[1] = l4[2] = l4[3] = ((unsigned)(i - t0) & (QT_TILE_REC_CAP - 1)) << 4;
    int hk = 0;
    for (int e = e0; e < e1; ++e) {
        const int cj = col[e];
        const float v = -(di * w[e] * dis[cj]);
        nrm[e] = v;
        const unsigned lj = (unsigned)(cj - t0);
        const bool remote = lj >= (unsigned)nr;
        int hs = 0;
        if (remote && nrem > 0) {
            hs = hbase + hk++;
            tile_halo[(int64_t)ts * QT_TILE_HALO_CAP + hs] = cj;
        }
        if (e - e0 < 4) {
            c4[e - e0] = cj;
            w4[e - e0] = v;
            l4[e - e0] = remote ? (((unsigned)hs << 4) | 1u) : ((lj & (QT_TILE_REC_CAP - 1)) << 4);
        } else if (run && e - e0 - 4 < 0xffff) {
            run[e - e0 - 4] = make_int2(remote ? (int)(0x80000000u | (unsigned)hs) : (int)lj, __float_as_int(v));
        }
    }
    if (ell) {
        if (e1 - e0 > 4) c4[3] = ~c4[3];
        ell[2 * (int64_t)i] = make_int4(c4[0], c4[1], c4[2], c4[3]);
        ell[2 * (int64_t)i + 1] = make_int4(__float_as_int(w4[0]), __float_as_int(w4[1]), __float_as_int(w4[2]), __float_as_int(w4[3]));
    }
    if (rec) {
        rec[0] = make_int4((int)(l4[0] | (l4[1] << 16)), (int)(l4[2] | (l4[3] << 16)), __float_as_int(w4[0]), __float_as_int(w4[1]));
        rec[1] = make_int4(__float_as_int(w4[2]), __float_as_int(w4[3]), (int)info, i - t0);
    }
}

}  // namespace

extern "C" int qt_edges_blocks(int N) { return N > 0 ? qt_cdiv(4 * (int64_t)N + 1, ET) : 0; }

extern "C" int qt_edges_count(const int32_t* labels, const int32_t* cell, int N, const int32_t* n_dev, int n, int m,
                              int32_t* cnt4, int32_t* sums, int32_t* tail_cnt, int B, int32_t* zero_buf, int zero_len, void* stream) {
    QT_ARG(labels && cell && cnt4 && sums, "null pointer");
    QT_ARG(!tail_cnt || B > 0, "tail_cnt needs the number of clips");
    QT_ARG(zero_len >= 0 && (zero_len == 0 || zero_buf), "zero_buf");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_edges_count, dim3(qt_edges_blocks(N)), dim3(ET), 0, (hipStream_t)stream, labels, cell, N, n_dev, n, m,
                       cnt4, sums, tail_cnt, B, zero_buf, zero_len);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_edges_fill(const int32_t* labels, const int32_t* cell, const int32_t* cnt4, const int32_t* sums, int N,
                             const int32_t* n_dev, int n, int m, float resolution, int32_t* rowptr, int32_t* col, float* w,
                             float* dis, void* stream) {
    QT_ARG(labels && cell && cnt4 && sums && rowptr && col && w && dis, "null pointer");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_edges_fill, dim3(qt_edges_blocks(N)), dim3(ET), 0, (hipStream_t)stream, labels, cell, cnt4, sums, N,
                       n_dev, n, m, resolution, rowptr, col, w, dis);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_tail_cap(void) { return QT_TAIL_CAP; }

extern "C" int qt_edges_norm(const int32_t* rowptr, const int32_t* col, const float* w, const float* dis, int N,
                             const int32_t* n_dev, float* nrm, int32_t* ell, const int32_t* cell, const int32_t* node_off,
                             int32_t* tail_cnt, int32_t* tail_pool, int32_t* tail_info, int32_t* tail_rec, void* stream) {
    QT_ARG(rowptr && col && w && dis && nrm, "null pointer");
    QT_ARG(((uintptr_t)ell & 15) == 0, "ell must be 16-byte aligned");
    QT_ARG(!tail_info || (cell && node_off && tail_cnt && tail_pool && ((uintptr_t)tail_pool & 15) == 0),
           "tail_info needs cell, node_off, tail_cnt (zeroed by qt_edges_count) and a 16-byte aligned tail_pool");
    QT_ARG(!tail_rec || (tail_info && ell && ((uintptr_t)tail_rec & 15) == 0), "tail_rec needs tail_info, ell and 16-byte alignment");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_edges_nrm, dim3(qt_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, w, dis, N, n_dev, nrm,
                       reinterpret_cast<int4*>(ell), cell, node_off, tail_cnt, reinterpret_cast<int2*>(tail_pool),
                       reinterpret_cast<uint32_t*>(tail_info), reinterpret_cast<int4*>(tail_rec));
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_edges_norm_tiles(const int32_t* rowptr, const int32_t* col, const float* w, const float* dis, int N,
                                   const int32_t* n_dev, float* nrm, int32_t* ell, const int32_t* cell, const int32_t* tile_off,
                                   int T, int nbj, int32_t* tile_cnt, int32_t* tile_pool, int32_t* tile_rec, int32_t* tile_brec,
                                   int32_t* tile_bpool, int32_t* tile_halo, int32_t* brec_addr, int32_t* err, void* stream) {
    QT_ARG(rowptr && col && w && dis && nrm && cell && tile_off && tile_cnt && tile_pool && tile_rec && tile_brec && tile_bpool && tile_halo &&
               brec_addr, "null pointer");
    QT_ARG(T >= 1 && nbj >= 1 && T % nbj == 0, "T tiles per clip in rows of nbj");
    QT_ARG((((uintptr_t)ell | (uintptr_t)tile_pool | (uintptr_t)tile_rec | (uintptr_t)tile_brec | (uintptr_t)tile_bpool) & 15) == 0,
           "ell and the tile arrays must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    hipLaunchKernelGGL(k_edges_nrm_tile, dim3(qt_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, w, dis, N, n_dev, nrm,
                       reinterpret_cast<int4*>(ell), cell, tile_off, T, nbj, tile_cnt, reinterpret_cast<int2*>(tile_pool),
                       reinterpret_cast<int4*>(tile_rec), reinterpret_cast<int4*>(tile_brec), reinterpret_cast<int2*>(tile_bpool),
                       tile_halo, brec_addr, err);
    QT_LAUNCHED();
    return QT_OK;
}
