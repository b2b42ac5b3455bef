// Clip-resident Chebyshev recurrences: ALL hops of one ChebConv pass in ONE launch, neighbour rows staged in LDS.
//
// The message aggregate  out = alpha L^ x + beta p + gamma q  (PyG ChebConv.propagate, model/model.py:53,96) is applied K - 1
// times in a row by every ChebConv stack: T_k = 2 L^ T_{k-1} - T_{k-2} forward, the Clenshaw recurrence backward.  As one
// launch per hop (k_spmm, cheb.hip) each hop is a grid-wide dependent chain index -> gather -> store through L2 / HBM, and
// at the benchmark's size (1.2e5 nodes) every launch is latency-bound at ~7.5 us for 24 MB.
//
// Two facts of the path make the whole recurrence local to a workgroup:
//   * the batched mesh is block diagonal -- a node's neighbours belong to its own clip, and a clip has at most n x m nodes
//     (4096 at 64 x 64);
//   * L^ acts on every channel separately, so a clip's (nodes x C) matrix splits into independent 4-channel column slices.
// One 1024-thread workgroup owns (clip c, slice s): a slice plane is 4096 rows x 16 B = 64 KB of LDS, two planes (T_{k-1}
// gathered, T_{k-2} read and overwritten in place by the owner of the row) are 128 KB of the CU's 160 KB.  The first four
// edges of each of a thread's four rows (the mesh build's ELL side array: column x4, weight x4) stay in registers for all
// hops, the edges beyond the fourth (rows of big cells beside small ones) come from a per-clip pool the mesh build wrote
// (qt_edges_norm: {local column, weight}, <= 4080 entries = the remaining 32 KB of LDS), the gathers are ds_read_b128, and
// between two hops there is one workgroup barrier -- waiting on LDS traffic only -- instead of a kernel boundary.
// HBM sees the operand once: Z read, K - 1 planes written (forward); K gradient planes read, one written (backward: the
// intermediate Clenshaw terms b_k never leave LDS).
//
// Arithmetic: the same fused multiply-adds in the same order as k_spmm (ELL slots 0..3, then the CSR tail of rows with more
// than four edges, then alpha * acc + beta * p + gamma * q), so the planes are bit-identical to the per-hop launches
// (tests/test_gpu_ops.py::test_clip_resident_recurrence_equals_per_hop_launches).
#include "qt_common.h"

namespace {

// Threads per workgroup and rows gathered together, measured at the bench shape (K = 5, C = 4 + 16, forward / backward, us per
// launch; tools/exp_clip.py): 1024 x 1 row: 21.4 / 21.3;  1024 x 2: 22.4 / 23.8;  512 x 2: 24.9 / 26.6;  512 x 4: 25.5 / 27.3 --
// sixteen waves that each wait on one row's five LDS reads beat eight waves with four rows in flight (per-hop launches: 29.6 / 31.4).
#ifndef QT_CLIP_T
#define QT_CLIP_T 1024
#endif
constexpr int CL_ROWS = QT_TAIL_REC_CAP;   // rows of a clip that fit: 2 planes x 4096 x 16 B = 128 KB
constexpr int CL_T = QT_CLIP_T;            // threads per workgroup
constexpr int CL_RPT = CL_ROWS / CL_T;     // rows per thread (6 registers per row for all hops: packed ELL columns, weights; the
                                           // backward 4 more for the prefetched A_k)
constexpr int CL_TAIL = QT_TAIL_CAP;       // LDS pool of tail edges (edges 5, 6, .. of a row) per clip: 8 B each, ~32 KB

struct ClipPart {
    const float* z;      // forward: T_0 slice source, (N, C) with row stride ld
    float* planes;       // forward: K - 1 output planes T_1 .., each SLICE-major (C / 4, Ncap, 4); backward: (K, Ncap, C) gradient planes
                         // (row-major), plane 0 rewritten
    int C, ld;
};

struct ClipArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* nrm;
    const int4* ell;
    const int32_t* node_off;       // (B + 1) first node of every clip (device): the valid rows of clip c are [off[c], off[c + 1])
    const int32_t* tail_cnt;       // (B * QT_TAIL_CNT_STRIDE): [0] tail edges of the clip in the pool (may exceed the capacity: see the
                                   // records' info word), [1] rows with a tail = records of the clip
    const int2* tail_pool;         // (B, CL_TAIL) {local column, weight bits}
    const int4* tail_rec;          // (B, CL_ROWS, 2 x int4) one record per row with more than four edges (qt_edges_norm):
                                   // {lc01, lc23, w0, w1}, {w2, w3, info, local row}; info = pool base | tail edges << 16, base 0xffff: walk the CSR
    int B, K, nsa;                 // nsa: slices of part a (part b's follow)
    int Ncap;                      // plane stride in rows (the capacity in static mode)
    int bwd_sm;                    // backward: the gradient planes 1 .. K-1 are slice-major (plane 0 is always row-major)
    ClipPart a, b;
    // TILE = true (frames of several 64 x 64 base cells; qt_cheb_tile_fwd / _bwd): a workgroup owns (clip, tile, slice)
    const int32_t* tile_off;       // (B T + 1) first node of every tile in label order
    const int32_t* tile_cnt;       // (B T * QT_TILE_CNT_STRIDE) counters of qt_edges_norm_tiles
    const int2* tile_pool;         // (B T, QT_TILE_POOL_CAP)
    const int4* tile_rec;          // (B T, QT_TILE_REC_CAP, 2 x int4) interior records
    const int4* tile_brec;         // (B T, QT_TILE_HALO_CAP, 2 x int4) boundary records
    const int2* tile_bpool;        // (B T, QT_TILE_BPOOL_CAP)
    const int32_t* tile_halo;      // (B T, QT_TILE_HALO_CAP) global row of every halo slot
    const int32_t* brec_addr;      // (N) boundary rows: tile slot * QT_TILE_HALO_CAP + boundary record index
    unsigned long long* xbuf;      // (B T, QT_TILE_SLICES, QT_TILE_HALO_CAP, 4) {value, tag} granules: the published boundary rows
    unsigned* sync;                // (2 B QT_TILE_SLICES): launch generation and arrivals per (clip, slice)
    unsigned* err;                 // the caller's persistent error word (never reset by the library): bit 0 a wait for a neighbour
                                   // tile gave up, bit 1 a tile capacity of the mesh build was exceeded
    int T, nbj, s0, ns;            // tiles per clip, tiles per tile row; first slice and slice count of THIS launch
#ifdef QT_CLIP_TIMING
    long long* dbg;                // diagnostics build (tools/exp_clip_timing.py): 16 stamps per workgroup
#endif
};
#ifdef QT_CLIP_TIMING
#define CL_STAMP(i) do { if (g.dbg && threadIdx.x == 0) g.dbg[(int64_t)blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define CL_STAMP(i) do {} while (0)
#endif

// the W channels of a slice row: a plain struct (independent registers: as one ext_vector value the 4-register tuples' alignment
// cost the W = 4 kernels ~25 more registers and spills); memory accesses go through the matching vector type
template <int W> struct fvec {
    float v[W];
    __device__ __forceinline__ float& operator[](int i) { return v[i]; }
    __device__ __forceinline__ const float& operator[](int i) const { return v[i]; }
};
template <int W> using fraw = float __attribute__((ext_vector_type(W)));
template <int W> __device__ __forceinline__ fvec<W> vload(const void* p) {
    const fraw<W> r = *reinterpret_cast<const fraw<W>*>(p);
    fvec<W> o;
#pragma unroll
    for (int i = 0; i < W; ++i) o.v[i] = r[i];
    return o;
}
template <int W> __device__ __forceinline__ void vstore(void* p, const fvec<W>& a) {
    fraw<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = a.v[i];
    *reinterpret_cast<fraw<W>*>(p) = r;
}
template <int W> __device__ __forceinline__ fvec<W> ldg(const float* p) { return vload<W>(p); }
template <int W> __device__ __forceinline__ fvec<W> vzero() {
    fvec<W> z;
#pragma unroll
    for (int i = 0; i < W; ++i) z.v[i] = 0.0f;
    return z;
}

// Workgroup barrier between two hops: only LDS is shared between the threads, so the barrier waits for this wave's LDS
// operations (lgkmcnt) and NOT for its global stores / prefetches (vmcnt), which __syncthreads() would also drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS rows are addressed by row * 16 (< 65536: two offsets per register); a slice row is 4 W bytes
template <int W> __device__ __forceinline__ fvec<W> lds_row(const char* plane, unsigned off16) {
    return vload<W>(plane + off16 * (unsigned)W / 4u);
}

// acc += w f, one fused multiply-add per channel.  Spelled with the intrinsic: left to the contraction pass, the vector form of
// w0 f0 + w1 f1 was fused the other way round in half the lanes -- 1 ulp away from k_spmm's chain ((0 + w0 f0) + w1 f1) + ...
template <int W> __device__ __forceinline__ void vfma(fvec<W>& a, float w, const fvec<W>& f) {
#pragma unroll
    for (int i = 0; i < W; ++i) a[i] = __builtin_fmaf(w, f[i], a[i]);
}

// The tail of a row with more than four edges, from the LDS copy of the clip's pool.  Four pool entries and their four
// gathers are in flight per trip; the accumulation order is the CSR order, as in k_spmm.
template <int W>
__device__ __forceinline__ void gather_tail_lds(fvec<W>& a, const char* __restrict__ P, const int2* __restrict__ TE, unsigned info) {
    const unsigned base = info & 0xffffu, cnt = info >> 16;
    for (unsigned j0 = 0; j0 < cnt; j0 += 4) {
        int2 e[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) e[v] = TE[base + min(j0 + v, cnt - 1)];
        fvec<W> f[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) f[v] = lds_row<W>(P, (unsigned)e[v].x << 4);
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (j0 + v < cnt) vfma<W>(a, __int_as_float(e[v].y), f[v]);
    }
}

// (pool full -- a clip with more than CL_TAIL tail edges: the row walks the CSR arrays instead; correct, slow, rare.  Forced
// by the small-caps test build: tests/test_gpu_ops.py::test_clip_resident_pool_overflow_walks_the_csr)
template <int W>
__device__ __forceinline__ void gather_tail_csr(fvec<W>& a, const char* __restrict__ P, unsigned row, int r0,
                                                const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                const float* __restrict__ nrm) {
    const int e0 = rowptr[row] + 4, e1 = rowptr[row + 1];
    for (int e = e0; e < e1; ++e) {
        const unsigned cj = (unsigned)((col[e] - r0) & (CL_ROWS - 1)) << 4;
        vfma<W>(a, nrm[e], lds_row<W>(P, cj));
    }
}

// W: channels per workgroup.  The hops are bound by the CU's own work (vector issue, LDS), not by memory: with W = 4 the
// benchmark's 32 clips x 4 .. 5 slices occupy 128 .. 160 of the 256 CUs; W = 2 (8-byte slice rows, twice the workgroups) is
// taken when B * C / 2 workgroups still fit the CUs in one round (clip_launch).  Same arithmetic per channel either way.
//
// TILE = true: frames of several 64 x 64 base cells.  A workgroup owns (clip, TILE, slice): a tile's nodes are one contiguous
// label range (qt_quadtree_stage3's cell_off) of at most 4096 rows, so its planes fit LDS exactly like a 64 x 64 clip's; what is
// new is that rows on the tile border have neighbours in other tiles of the clip.  Per tile the mesh build (qt_edges_norm_tiles)
// leaves: boundary records (rows with an edge that leaves the tile: first four edges as local rows or HALO slots, the rest in a
// boundary pool) and the halo list (the global row behind every slot; <= 256 per tile).  Interior rows run exactly as in the
// one-tile kernel.  After every hop the T workgroups of a (clip, slice) exchange their boundary rows THROUGH GLOBAL MEMORY:
//   hop c:  waves 0-3 finish the boundary rows first (their neighbours: LDS plane or the halo buffer HB = T_{c-1} of the halo
//           rows) and PUBLISH each as four 8-byte granules {value, tag} in the tile's exchange slots (xbuf; two 16-byte write-
//           through stores of two granules each), tag =
//           launch generation of the (clip, slice) * 16 + c;  after their interior rows they poll the granules of THEIR halo
//           row (sc1 loads from the owner tile's slots: brec_addr of the row) until all four tags match and write HB before
//           the hop's one workgroup barrier.  (Requesting the granules half way through the interior rows gained nothing: a
//           publication takes ~1.5 us to become visible and a poll ~1.5 us to return, so the early poll usually missed.)
//   The data is its own flag (MI355X_MICROARCH.md, hand-off form R2: naturally aligned {value, tag} granules, observed untorn
//   also as the two halves of a 16-byte sc1 access; "granules for latency"): no drain, no separate flag, no dependent load
//   behind a poll; a reader that catches a row between its two stores sees a stale tag in one half and polls again.  A first version with write-through rows + a
//   drained per-tile hop flag + sc1 halo loads behind the poll took 6 - 7 us per hop -- no faster than a k_spmm launch per hop
//   (8 us at 128 x 128 x 8 clips); with granules a hop costs ~4.3 us (tools/exp_tile.py: K = 5 at 128 x 128 x 8 clips 30 -> 21 us
//   forward, 32 -> 27 us backward; 32 channels 38 -> 25 / 46 -> 32), which is the chain boundary rows (0.6 us) -> store visible
//   (~1.5 us) -> poll round trip (~1.5 us): what a cross-CU hand-off costs on this chip, tile size notwithstanding.  So the path
//   pays for K >= 4 when all workgroups fit the CUs in one round, and is NOT taken otherwise (ops._tile_resident): two hops are
//   no faster than two k_spmm launches, and 640 workgroups (hidden 32: 16 clips x 4 tiles x 10 slices) take three rounds.
//   Tags only ever grow (the generation word is bumped by the last workgroup of the (clip, slice) to leave), so nothing is reset
//   between launches or hipGraph replays; the exchange buffer is zeroed once per mesh build.
// No dependence on dispatch order beyond forward progress: spins are bounded (a timeout sets the error word and the launch
// finishes with garbage instead of hanging) and the launches are cut so that all workgroups of one are co-resident.
#ifndef QT_TILE_SPIN_LIMIT
#define QT_TILE_SPIN_LIMIT 200000u                         // polls of ~1 us: ~0.2 s before a poll gives up (a test build shrinks it)
#endif
constexpr unsigned TILE_SPIN_LIMIT = QT_TILE_SPIN_LIMIT;
static_assert((QT_TILE_HALO_CAP & (QT_TILE_HALO_CAP - 1)) == 0 && QT_TILE_HALO_CAP <= 256,
              "halo slots: a power of two (slot masks) of at most 256 (the boundary pool keeps a slot in 8 bits; waves 0-3 own them)");

template <bool BWD, int W, bool TILE = false>
__global__ __launch_bounds__(CL_T) void k_cheb_clip(ClipArgs g) {
    using V = fvec<W>;
    static_assert(!TILE || W == 4, "the tile exchange moves 16-byte rows");
    constexpr int NPOOL = TILE ? QT_TILE_POOL_CAP : CL_TAIL;
    __shared__ __attribute__((aligned(16))) char Pl[2 * CL_ROWS * 4 * W];
    __shared__ int2 TE[NPOOL];
    __shared__ __attribute__((aligned(16))) char HB[TILE ? QT_TILE_HALO_CAP * 4 * W : 16];     // T_{c-1} of the halo rows
    __shared__ int4 BR[TILE ? 2 * QT_TILE_HALO_CAP : 1];                                        // boundary records
    __shared__ int2 BP[TILE ? QT_TILE_BPOOL_CAP : 1];                                           // boundary pool
    __shared__ int HC[TILE ? QT_TILE_HALO_CAP : 1];                                             // granule index of every halo slot's row in xbuf
    __shared__ unsigned arrive;                                                                 // boundary waves through with the halo buffer, 4 per hop
    const int t = threadIdx.x;
    int c, s, ts = 0, tl = 0;
    if constexpr (TILE) {
        // groups (clip, slice) of this launch; the T workgroups of a group get ids 8 apart (one XCD under round-robin
        // placement: a speed matter only) inside a window of 8 T consecutive ids
        const int gi = ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x / (8 * g.T));
        tl = ((int)blockIdx.x >> 3) % g.T;
        if (gi >= g.B * g.ns) return;
        c = gi % g.B;
        s = g.s0 + gi / g.B;
        ts = c * g.T + tl;
    } else {
        c = (int)blockIdx.x % g.B;
        s = (int)blockIdx.x / g.B;
    }
    const bool second = s >= g.nsa;
    const ClipPart& pt = second ? g.b : g.a;
    const int C = pt.C;
    const int ch = W * (second ? s - g.nsa : s);
    const int r0 = TILE ? g.tile_off[ts] : g.node_off[c];
    const int nr = min((TILE ? g.tile_off[ts + 1] : g.node_off[c + 1]) - r0, CL_ROWS);
    // TILE: the sync words of this (clip, slice): launch generation (read by every workgroup at its start, bumped by the last
    // one to leave) and the count of workgroups that have left
    unsigned* const ggen = TILE ? g.sync + ((int64_t)c * QT_TILE_SLICES + s) : nullptr;
    unsigned* const gdone = TILE ? g.sync + ((int64_t)(g.B + c) * QT_TILE_SLICES + s) : nullptr;
    unsigned* const gerr = TILE ? g.err : nullptr;
    unsigned tag0 = 0;                       // generation * 16: a hop's tag is tag0 + hop number
    if constexpr (TILE) tag0 = __hip_atomic_load(ggen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 4;
    auto leave = [&]() {
        if constexpr (TILE) {
            if (t == 0) {
                const unsigned old = __hip_atomic_fetch_add(gdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == (unsigned)g.T - 1u) {          // every workgroup of the group has read the generation and left
                    __hip_atomic_store(gdone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(ggen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    if (nr <= 0) {                                         // (workgroup-uniform)
        leave();
        return;
    }
    const int32_t* cnts = TILE ? g.tile_cnt + (int64_t)QT_TILE_CNT_STRIDE * ts : g.tail_cnt + QT_TAIL_CNT_STRIDE * c;
    const int ntail = min(cnts[0], NPOOL);
    const int ntr = min(cnts[1], CL_ROWS);
    const int nh = TILE ? min(cnts[2], QT_TILE_HALO_CAP) : 0;
    const int nb = TILE ? min(cnts[3], QT_TILE_HALO_CAP) : 0;
    const int nbp = TILE ? min(cnts[4], QT_TILE_BPOOL_CAP) : 0;
    if constexpr (TILE) {
        if (t == 0) {
            arrive = 0u;
            if (cnts[5]) __hip_atomic_fetch_or(gerr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a tile capacity was exceeded)
        }
    }
    bool dead = false;                       // TILE: a poll timed out: no further waiting, the error word is set
    CL_STAMP(0);
    const int K = g.K;
    const unsigned pstride = (unsigned)g.Ncap * (unsigned)C;      // (K * Ncap * C < 2^31: checked by the host entry -- 32-bit offsets)

    // element offset of this slice's channels of (gradient plane k, row): row-major (K, Ncap, C), or planes 1.. slice-major
    auto grad_off = [&](int k, unsigned row) -> unsigned {
        if (g.bwd_sm && k > 0) return (unsigned)k * pstride + ((unsigned)(ch >> 2) * (unsigned)g.Ncap + row) * 4u + (unsigned)(ch & 3);
        return (unsigned)k * pstride + row * (unsigned)C + ch;
    };
    // Rows with MORE than four edges (big cells beside small ones: 3 - 4 % of the rows of the benchmark's meshes) are not
    // finished by the thread that owns them.  Handled in place -- every thread checking its rows for a tail -- nearly every
    // wave walked the tail loop in every one of its row trips for one or two active lanes: 10 of 24 us of a K = 5 launch.
    // The mesh build leaves the clip's tail rows as a compact list of records instead, and record j goes to the row SLOT nr + j:
    // the slots past the clip's nr rows are idle anyway, and there are enough of them -- a row with a tail is a cell of at
    // least four pixels (a one-pixel cell has at most four neighbours), which frees three slots of the n x m <= 4096.  The
    // thread that owns slot nr + j runs that row entirely (first four edges from the record, then the pool entries, then the
    // row's finish), so the tail loop runs in the two or three waves that hold records, with every lane busy; the row's own
    // thread still writes the first plane and gathers, but stores nothing.  Every record finds a slot: rows + rows with a tail
    // <= pixels <= CL_ROWS (tests/test_gpu_mesh.py::test_tail_row_records checks it on every mesh it builds; round 3 carried a
    // per-hop re-read loop for slot-less records that no mesh could reach -- removed).
    const int nslot = min(ntr, CL_ROWS - nr);
    // Prologue, ONE memory phase: the first four edges of this thread's rows (kept in registers for every hop) or the records
    // of its slots, the rows' first operand and the clip's tail pool -- all requested before anything is used.  Rows past the
    // clip's count are clamped to its last row (valid loads, results discarded), so no load sits behind a branch.
    unsigned rowc[CL_RPT];
    int4 c4[CL_RPT], wb[CL_RPT];
    V first[CL_RPT];
    V nxt[CL_RPT];
#pragma unroll
    for (int u = 0; u < CL_RPT; ++u) {
        const int i = t + CL_T * u;
        rowc[u] = (unsigned)(r0 + min(i, nr - 1));
        const bool isrec = i >= nr && i - nr < nslot;
        const int4* src = isrec ? (TILE ? g.tile_rec + 2 * ((int64_t)ts * QT_TILE_REC_CAP + (i - nr))
                                        : g.tail_rec + 2 * ((int64_t)c * CL_ROWS + (i - nr)))
                                : g.ell + 2 * (int64_t)rowc[u];
        c4[u] = src[0];
        wb[u] = src[1];
        if constexpr (!BWD) {
#ifdef QT_EXP_CLIP_SMZ            // (timing experiment: the first operand read as if it were stored slice-major -- wrong values, same buffer)
            first[u] = ldg<W>(pt.z + (((unsigned)(ch >> 2) * (unsigned)g.Ncap + rowc[u]) * 4u + (unsigned)(ch & 3)));
#else
            first[u] = ldg<W>(pt.z + (rowc[u] * (unsigned)pt.ld + ch));
#endif
        } else {
            first[u] = ldg<W>(pt.planes + grad_off(K - 1, rowc[u]));
        }
    }
    {
        const int2* src = TILE ? g.tile_pool + (int64_t)ts * QT_TILE_POOL_CAP : g.tail_pool + (int64_t)c * CL_TAIL;
        int2 te[(NPOOL + CL_T - 1) / CL_T];
#pragma unroll
        for (int i = 0; i < (NPOOL + CL_T - 1) / CL_T; ++i)
            if (t + CL_T * i < ntail) te[i] = src[t + CL_T * i];
#pragma unroll
        for (int i = 0; i < (NPOOL + CL_T - 1) / CL_T; ++i)
            if (t + CL_T * i < ntail) TE[t + CL_T * i] = te[i];
    }
    // TILE: the boundary records, their pool and the halo list of this tile -> LDS; the halo rows' first operand -> HB (an
    // input of the launch: plain loads, nothing to wait for)
    V bnx = vzero<W>();                      // backward: A_k of this thread's boundary record row, requested one hop ahead
    if constexpr (TILE) {
        if (t < nb) {
            const int4* bs = g.tile_brec + 2 * ((int64_t)ts * QT_TILE_HALO_CAP + t);
            const int4 b0 = bs[0], b1 = bs[1];
            BR[2 * t] = b0;
            BR[2 * t + 1] = b1;
            if constexpr (BWD) bnx = ldg<W>(pt.planes + grad_off(K - 2, (unsigned)r0 + ((unsigned)b1.w & (CL_ROWS - 1))));
        }
        if (t < nbp) BP[t] = g.tile_bpool[(int64_t)ts * QT_TILE_BPOOL_CAP + t];
        if (t < nh) {
            const int hc = g.tile_halo[(int64_t)ts * QT_TILE_HALO_CAP + t];
            // granule index of the halo row's published values: (owner tile * slices + this slice) * 256 + its record, * 4
            // (negative: the owner tile's boundary record did not fit -- qt_edges_norm_tiles reported it; nobody publishes that
            // row, so this lane never waits for it)
            const int ba = g.brec_addr[hc];
            if (ba < 0) {
                dead = true;
                __hip_atomic_fetch_or(gerr, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            HC[t] = ba < 0 ? 0 : (((ba / QT_TILE_HALO_CAP) * QT_TILE_SLICES + s) * 2 * QT_TILE_HALO_CAP + (ba & (QT_TILE_HALO_CAP - 1))) * 4;
            V hv;
            if constexpr (!BWD) hv = ldg<W>(pt.z + ((unsigned)hc * (unsigned)pt.ld + ch));
            else hv = ldg<W>(pt.planes + grad_off(K - 1, (unsigned)hc));
            vstore<W>(HB + t * (4 * W), hv);
        }
    }
    CL_STAMP(1);
    unsigned lc[CL_RPT][2];
    float w[CL_RPT][4];
    unsigned tinfo[CL_RPT];                 // record slots: pool base | tail edge count << 16 (base 0xffff: walk the CSR); 0 otherwise
    unsigned mine = 0;                      // bit u: this thread finishes row u (its own row without a tail, or a record's row)
    bool tails = false;
#pragma unroll
    for (int u = 0; u < CL_RPT; ++u) {
        const int i = t + CL_T * u;
        const bool ok = i < nr;
        if (ok) vstore<W>(Pl + i * (4 * W), first[u]);
        if (i >= nr && i - nr < nslot) {    // a record: {lc01, lc23, w0, w1}, {w2, w3, info, local row}
            lc[u][0] = (unsigned)c4[u].x;
            lc[u][1] = (unsigned)c4[u].y;
            w[u][0] = __int_as_float(c4[u].z);
            w[u][1] = __int_as_float(c4[u].w);
            w[u][2] = __int_as_float(wb[u].x);
            w[u][3] = __int_as_float(wb[u].y);
            tinfo[u] = (unsigned)wb[u].z;
            rowc[u] = (unsigned)r0 + ((unsigned)wb[u].w & (CL_ROWS - 1));
            mine |= 1u << u;
            tails = true;
        } else {
            const bool tail = c4[u].w < 0;  // (the ELL array flags a row with more edges by complementing its last column)
            if (tail) c4[u].w = ~c4[u].w;
            bool remote = false;            // TILE: a neighbour in another tile -> the row runs from its boundary record
            if constexpr (TILE)
                remote = (unsigned)(c4[u].x - r0) >= (unsigned)nr || (unsigned)(c4[u].y - r0) >= (unsigned)nr ||
                         (unsigned)(c4[u].z - r0) >= (unsigned)nr || (unsigned)(c4[u].w - r0) >= (unsigned)nr;
            if (ok && !tail && !remote) mine |= 1u << u;
            lc[u][0] = ((unsigned)((c4[u].x - r0) & (CL_ROWS - 1)) << 4) | ((unsigned)((c4[u].y - r0) & (CL_ROWS - 1)) << 20);
            lc[u][1] = ((unsigned)((c4[u].z - r0) & (CL_ROWS - 1)) << 4) | ((unsigned)((c4[u].w - r0) & (CL_ROWS - 1)) << 20);
            w[u][0] = __int_as_float(wb[u].x);
            w[u][1] = __int_as_float(wb[u].y);
            w[u][2] = __int_as_float(wb[u].z);
            w[u][3] = __int_as_float(wb[u].w);
            tinfo[u] = 0;
        }
    }
    CL_STAMP(2);
    if constexpr (BWD) {       // A_{K-2} of the rows: needed at the end of the first hop's groups (requested here, not in the prologue:
#pragma unroll                 // its registers would sit beside the ELL vectors' and spill)
        for (int u = 0; u < CL_RPT; ++u) nxt[u] = ldg<W>(pt.planes + grad_off(K - 2, rowc[u]));
    }
    lds_barrier();
    CL_STAMP(3);
    int stamp = 4;
    (void)stamp;
    // One hop with the gathered plane at byte offset CO of Pl (compile-time: the hop loops below are unrolled by two, so the plane
    // offsets are instruction immediates).  Per row: its four gathers AND its own old value (OWN: the plane being overwritten) are
    // requested together -- one LDS round trip (the first version took two per row: 3.6 us per hop with 0.7 us of gather work
    // in it) -- then the tail of a record's row,  r = alpha * acc;  addend(r) (backward: + A_k of the row);  r += beta * own;
    // fin(row, r) stores the row.
    constexpr unsigned PLANE = (unsigned)CL_ROWS * 4u * W;
    // TILE: hopc = 1, 2, .. counts the hops of the launch (the tag of its granules); more = a further hop follows (the last hop
    // publishes and fetches nothing)
    struct Xch { unsigned hopc; bool more; };
    // (num_records = the exchange buffer's real size: an out-of-range poll reads 0, never matches a tag and times out)
    [[maybe_unused]] const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        TILE ? g.xbuf : nullptr, 0, TILE ? g.B * g.T * (QT_TILE_SLICES * 2 * QT_TILE_HALO_CAP * 32) : 0, 0x00020000);
    using U4 = unsigned __attribute__((ext_vector_type(4)));
    auto hop = [&](auto co_tag, auto own_tag, float alpha, float beta, auto&& addend, auto&& fin, const Xch& x) {
        constexpr unsigned CO = decltype(co_tag)::value;
        constexpr bool OWN = decltype(own_tag)::value;
        const char* Pc = Pl + CO;
        char* Pn = Pl + (CO ^ PLANE);
        auto finish = [&](const V& acc, const V& own, V& nx, char* slot, unsigned grow, bool live, auto&& fn) {
            V r;
#pragma unroll
            for (int i = 0; i < W; ++i) r[i] = alpha * acc[i];
            addend(nx, grow, r);                               // (backward: + A_k of the row)
            if constexpr (OWN) {
#pragma unroll
                for (int i = 0; i < W; ++i) r[i] = __builtin_fmaf(beta, own[i], r[i]);
            }
            if (live) fn(grow, r, slot);
        };
        [[maybe_unused]] const unsigned tag = tag0 + x.hopc;
        if constexpr (TILE) {
            // ---- boundary rows first (waves 0-3, one record per thread): neighbours from the LDS plane or the halo buffer, the
            // same chain of fused multiply-adds in CSR order; the row is stored like any other and, when a further hop follows,
            // PUBLISHED as four {value, tag} granules in this tile's exchange slots (two 16-byte write-through stores)
            if (t < nb) {
                const int4 b0 = BR[2 * t], b1 = BR[2 * t + 1];
                const unsigned lrow = (unsigned)b1.w & (CL_ROWS - 1), info = (unsigned)b1.z;
                auto gat = [&](unsigned f16) -> V {
                    return (f16 & 1u) ? vload<W>(HB + (f16 >> 4) * (4u * W)) : lds_row<W>(Pc, f16 & 0xfff0u);
                };
                V f[4], own = vzero<W>();
                f[0] = gat((unsigned)b0.x & 0xffffu);
                f[1] = gat((unsigned)b0.x >> 16);
                f[2] = gat((unsigned)b0.y & 0xffffu);
                f[3] = gat((unsigned)b0.y >> 16);
                char* slot = Pn + lrow * (4u * W);
                if constexpr (OWN) own = vload<W>(slot);
                const float bw[4] = {__int_as_float(b0.z), __int_as_float(b0.w), __int_as_float(b1.x), __int_as_float(b1.y)};
                V acc = vzero<W>();
#pragma unroll
                for (int e = 0; e < 4; ++e) vfma<W>(acc, bw[e], f[e]);
                const bool real = info != 0xffffffffu;          // (a NULL record: a row whose record did not fit, qt_edges_norm_tiles)
                const unsigned pb = info & 0xffffu, pc = real ? info >> 16 : 0u;
                for (unsigned j0 = 0; j0 < pc; j0 += 4) {
                    int2 e[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) e[v] = BP[pb + min(j0 + v, pc - 1)];
                    V ff[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        ff[v] = e[v].x < 0 ? vload<W>(HB + ((unsigned)e[v].x & 0xffu) * (4u * W)) : lds_row<W>(Pc, (unsigned)e[v].x << 4);
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (j0 + v < pc) vfma<W>(acc, __int_as_float(e[v].y), ff[v]);
                }
                finish(acc, own, bnx, slot, (unsigned)r0 + lrow, real, [&](unsigned grow, const V& r, char* own_slot) {
                    fin(grow, r, own_slot);
                    if (x.more) {
                        const int off = (((ts * QT_TILE_SLICES + s) * 2 + (int)(x.hopc & 1u)) * QT_TILE_HALO_CAP + t) * 32;
                        const U4 g0 = {__float_as_uint(r[0]), tag, __float_as_uint(r[1]), tag};
                        const U4 g1 = {__float_as_uint(r[2]), tag, __float_as_uint(r[3]), tag};
                        __builtin_amdgcn_raw_buffer_store_b128(g0, xrsrc, off, 0, 16);           // sc1
                        __builtin_amdgcn_raw_buffer_store_b128(g1, xrsrc, off + 16, 0, 16);
                    }
                });
            }
            // (the halo buffer may be overwritten once all four boundary waves are through with it: they count themselves here)
            if (x.more && t < 256 && (t & 63) == 0) atomicAdd(&arrive, 1u);
        }
        // (the packed column offsets and the row numbers are made opaque once per hop: otherwise the loop-invariant unpacked LDS
        // addresses -- one per gather -- and 64-bit row addresses are kept in registers across the hops, and the kernel spills)
#pragma unroll
        for (int u = 0; u < CL_RPT; ++u) asm volatile("" : "+v"(lc[u][0]), "+v"(lc[u][1]), "+v"(rowc[u]));
        [[maybe_unused]] const int xoff = TILE ? (HC[t < nh ? t : 0] + (int)(x.hopc & 1u) * QT_TILE_HALO_CAP * 4) * 8 : 0;
#pragma unroll
        for (int u = 0; u < CL_RPT; ++u) {
            V f[4], own = vzero<W>();
            f[0] = lds_row<W>(Pc, lc[u][0] & 0xffffu);
            f[1] = lds_row<W>(Pc, lc[u][0] >> 16);
            f[2] = lds_row<W>(Pc, lc[u][1] & 0xffffu);
            f[3] = lds_row<W>(Pc, lc[u][1] >> 16);
            char* slot = Pn + (rowc[u] - (unsigned)r0) * (4u * W);
            if constexpr (OWN) own = vload<W>(slot);
            V acc = vzero<W>();                                // (the same chain of fused multiply-adds as k_spmm)
#pragma unroll
            for (int e = 0; e < 4; ++e) vfma<W>(acc, w[u][e], f[e]);
            if (tails && tinfo[u]) {
                if ((tinfo[u] & 0xffffu) != 0xffffu)
                    gather_tail_lds<W>(acc, Pc, TE, tinfo[u]);
                else
                    gather_tail_csr<W>(acc, Pc, rowc[u], r0, g.rowptr, g.col, g.nrm);
            }
            finish(acc, own, nxt[u], slot, rowc[u], (mine >> u) & 1u, fin);      // (loads unconditional, the stores predicated)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (TILE) {
            if (x.more && t < 256) {
                // this thread's halo row as its owner tile published it in THIS hop: poll the four granules until every tag says so
                // (each wave for itself; the owners published before their interior rows, so the first poll usually matches)
                V hv = vzero<W>();
                if (!dead) {
                    unsigned spins = 0;
                    for (;;) {
                        bool ok = true;
                        if (t < nh) {
                            const U4 q0 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff, 0, 16);          // sc1
                            const U4 q1 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xoff + 16, 0, 16);
                            ok = q0.y == tag && q0.w == tag && q1.y == tag && q1.w == tag;
                            hv[0] = __uint_as_float(q0.x);
                            hv[1] = __uint_as_float(q0.z);
                            hv[2] = __uint_as_float(q1.x);
                            hv[3] = __uint_as_float(q1.z);
                        }
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
                        if (++spins > TILE_SPIN_LIMIT) {
                            dead = true;
                            if ((t & 63) == 0) __hip_atomic_fetch_or(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                while (*(volatile unsigned*)&arrive < 4u * x.hopc) __builtin_amdgcn_s_sleep(1);
                if (t < nh) vstore<W>(HB + t * (4 * W), hv);
            }
        }
        lds_barrier();
        CL_STAMP(stamp++);
    };
    using Co0 = std::integral_constant<unsigned, 0u>;
    using Co1 = std::integral_constant<unsigned, PLANE>;
    using Yes = std::true_type;
    using No = std::false_type;
    if constexpr (!BWD) {
        // T_1 = L^ T_0;  T_k = 2 L^ T_{k-1} - T_{k-2}: the owner of a row reads its old value and overwrites it (nobody else reads
        // that plane during the hop)
        auto none = [](V&, unsigned, V&) {};
        auto store = [&](int k) {
            // planes leave SLICE-major, (plane, 4-channel slice, N, 4): this workgroup's rows are contiguous, its stores coalesce
            // (row-major planes: 16-byte pieces 4 C bytes apart, 21.3 vs 19.5 us per K = 5 launch); the GEMMs that read the
            // planes reach a row's quad at slice base + 4 row (PlaneSrc.sm)
            float* outp = pt.planes + ((unsigned)(k - 1) * pstride + (unsigned)(ch >> 2) * (unsigned)g.Ncap * 4u + (unsigned)(ch & 3));
            return [=](unsigned grow, const V& r, char* own) {
                vstore<W>(own, r);
#ifndef QT_EXP_CLIP_NOSTORE         // (timing experiment)
                vstore<W>(outp + grow * 4u, r);
#endif
            };
        };
        auto xch = [&](int k) { return Xch{(unsigned)k, k + 1 < K}; };
        hop(Co0{}, No{}, 1.0f, 0.0f, none, store(1), xch(1));
        for (int k = 2; k < K; k += 2) {
            hop(Co1{}, Yes{}, 2.0f, -1.0f, none, store(k), xch(k));
            if (k + 1 < K) hop(Co0{}, Yes{}, 2.0f, -1.0f, none, store(k + 1), xch(k + 1));
        }
    } else {
        // Clenshaw on the gradient planes A_0 .. A_{K-1}: b_{K-1} = A_{K-1}; b_k = A_k + 2 L^ b_{k+1} - b_{k+2};
        // out = A_0 + L^ b_1 - b_2 (written over A_0).  A_k is the row's own entry of plane k, requested during the hop before.
        struct AddAk {
            const float* planes;
            decltype(grad_off)& off;
            int k;
            __device__ __forceinline__ void operator()(V& nx, unsigned grow, V& r) const {
                const V ak = nx;                              // A_k of this row; its A_{k-1} is requested as soon as A_k is consumed
#pragma unroll
                for (int i = 0; i < W; ++i) r[i] += 1.0f * ak[i];
                if (k > 0) nx = ldg<W>(planes + off(k - 1, grow));
            }
        };
        auto add_ak = [&](int k) { return AddAk{pt.planes, grad_off, k}; };
        // b_k stays in LDS; the last hop (k = 0) writes the result over A_0 in global memory
        auto put = [&](int k) {
            return [=](unsigned grow, const V& r, char* own) {
                if (k == 0)
                    vstore<W>(pt.planes + (grow * (unsigned)C + ch), r);
                else
                    vstore<W>(own, r);
            };
        };
        auto xch = [&](int k) { return Xch{(unsigned)(K - 1 - k), k > 0}; };
        int k = K - 2;
        hop(Co0{}, No{}, k == 0 ? 1.0f : 2.0f, 0.0f, add_ak(k), put(k), xch(k));          // b_{K-2} = A_{K-2} + 2 L^ b_{K-1}
        for (--k; k >= 0; k -= 2) {
            hop(Co1{}, Yes{}, k == 0 ? 1.0f : 2.0f, -1.0f, add_ak(k), put(k), xch(k));
            if (k - 1 >= 0) hop(Co0{}, Yes{}, k - 1 == 0 ? 1.0f : 2.0f, -1.0f, add_ak(k - 1), put(k - 1), xch(k - 1));
        }
    }
    leave();
}

}  // namespace

extern "C" int qt_cheb_clip_rows(void) { return CL_ROWS; }
#ifdef QT_CLIP_TIMING
static long long* g_clip_dbg = nullptr;
extern "C" void qt_clip_timing_buffer(long long* p) { g_clip_dbg = p; }
#endif

struct ClipMesh {       // the mesh operands both entry points share
    const int32_t *rowptr, *col;
    const float* nrm;
    const int32_t *ell, *node_off, *tail_cnt, *tail_pool, *tail_rec;
    int B;
};

// Slice width (the `width` argument of both entry points): 0 = automatic (2 when the launch's B * C / 2 workgroups fit the CUs in
// one round, else 4); 2 or 4 pins it (diagnostics and the parity tests of both widths).  No library-global switch: the ABI keeps
// no mutable state besides the thread-local error string.
extern "C" int qt_num_cus(void);

static int clip_launch(bool bwd, const ClipMesh& m, int Ncap, int K, int Ca, const float* za, int lda, float* Pa, int Cb,
                       const float* zb, int ldb, float* Pb, void* stream, int width, int bwd_sm = 0) {
    // forward: half-width slices whenever they still fit the CUs in one round (K = 5, 16 channels, 32 clips: 16.8 -> 15.2 us);
    // backward only when even they leave half the CUs idle: its A_k loads and the final store move 8 of every 16 bytes at
    // W = 2 (the same shape backward: 19.1 us at W = 4, 21.0 at W = 2; 4 channels: 13.2 vs 11.5)
    const int half = m.B * ((Ca + Cb) / 2);
    const int W = width ? width : (half <= (bwd ? qt_num_cus() / 2 : qt_num_cus()) ? 2 : 4);
    ClipArgs g;
    g.bwd_sm = bwd_sm != 0;
    g.rowptr = m.rowptr;
    g.col = m.col;
    g.nrm = m.nrm;
    g.ell = reinterpret_cast<const int4*>(m.ell);
    g.node_off = m.node_off;
    g.tail_cnt = m.tail_cnt;
    g.tail_pool = reinterpret_cast<const int2*>(m.tail_pool);
    g.tail_rec = reinterpret_cast<const int4*>(m.tail_rec);
    g.B = m.B;
    g.K = K;
    g.nsa = Ca / W;
    g.Ncap = Ncap;
#ifdef QT_CLIP_TIMING
    g.dbg = g_clip_dbg;
#endif
    g.a = ClipPart{za, Pa, Ca, lda ? lda : Ca};
    g.b = ClipPart{zb, Pb, Cb, ldb ? ldb : Cb};
    const int grid = m.B * ((Ca + Cb) / W);
    if (bwd && W == 2)
        hipLaunchKernelGGL((k_cheb_clip<true, 2>), dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    else if (bwd)
        hipLaunchKernelGGL((k_cheb_clip<true, 4>), dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    else if (W == 2)
        hipLaunchKernelGGL((k_cheb_clip<false, 2>), dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    else
        hipLaunchKernelGGL((k_cheb_clip<false, 4>), dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    return 0;
}

#define CLIP_MESH_ARGS_OK                                                                                                        \
    QT_ARG(rowptr && col && nrm && ell && node_off && tail_cnt && tail_pool && tail_rec && B > 0 && K >= 2,                     \
           "bad arguments (the ELL side array, the tail pool and the tail-row records of qt_edges_norm are required)")

extern "C" int qt_cheb_clip_fwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* node_off, const int32_t* tail_cnt, const int32_t* tail_pool,
                                const int32_t* tail_rec, int B, int N, int K, int Ca, const float* za, int lda, float* Ta, int Cb,
                                const float* zb, int ldb, float* Tb, int width, void* stream) {
    CLIP_MESH_ARGS_OK;
    QT_ARG(width == 0 || width == 2 || width == 4, "width must be 0 (automatic), 2 or 4");
    QT_ARG(za && Ta && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || (zb && Tb)), "bad operands");
    QT_ARG((lda | ldb) % 4 == 0, "row strides must be multiples of 4");
    QT_ARG((((uintptr_t)za | (uintptr_t)Ta | (uintptr_t)zb | (uintptr_t)Tb | (uintptr_t)ell | (uintptr_t)tail_pool | (uintptr_t)tail_rec) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31) && (int64_t)N * max(max(lda, ldb), 4) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    const ClipMesh m = {rowptr, col, nrm, ell, node_off, tail_cnt, tail_pool, tail_rec, B};
    clip_launch(false, m, N, K, Ca, za, lda, Ta, Cb, zb, ldb, Tb, stream, width);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_cheb_clip_bwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* node_off, const int32_t* tail_cnt, const int32_t* tail_pool,
                                const int32_t* tail_rec, int B, int N, int K, int Ca, float* Ga, int Cb, float* Gb, int planes_sm,
                                int width, void* stream) {
    CLIP_MESH_ARGS_OK;
    QT_ARG(width == 0 || width == 2 || width == 4, "width must be 0 (automatic), 2 or 4");
    QT_ARG(Ga && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || Gb), "bad operands");
    QT_ARG((((uintptr_t)Ga | (uintptr_t)Gb | (uintptr_t)ell | (uintptr_t)tail_pool | (uintptr_t)tail_rec) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    const ClipMesh m = {rowptr, col, nrm, ell, node_off, tail_cnt, tail_pool, tail_rec, B};
    clip_launch(true, m, N, K, Ca, nullptr, 0, Ga, Cb, nullptr, 0, Gb, stream, width, planes_sm);
    QT_LAUNCHED();
    return QT_OK;
}

// ---- frames of several 64 x 64 base cells: the same recurrences with one workgroup per (clip, TILE, slice) and a halo exchange
// between the tiles of a clip after every hop (k_cheb_clip<.., TILE = true>)
struct TileMesh {
    const int32_t *rowptr, *col;
    const float* nrm;
    const int32_t *ell, *tile_off, *tile_cnt, *tile_pool, *tile_rec, *tile_brec, *tile_bpool, *tile_halo, *brec_addr;
    int32_t *xbuf, *sync, *err;
    int B, T, nbj;
};

static int tile_launch(bool bwd, const TileMesh& m, int Ncap, int K, int Ca, const float* za, int lda, float* Pa, int Cb,
                       const float* zb, int ldb, float* Pb, void* stream, int bwd_sm) {
    ClipArgs g;
    g.bwd_sm = bwd_sm != 0;
    g.rowptr = m.rowptr;
    g.col = m.col;
    g.nrm = m.nrm;
    g.ell = reinterpret_cast<const int4*>(m.ell);
    g.node_off = nullptr;
    g.tail_cnt = nullptr;
    g.tail_pool = nullptr;
    g.tail_rec = nullptr;
    g.B = m.B;
    g.K = K;
    g.nsa = Ca / 4;
    g.Ncap = Ncap;
#ifdef QT_CLIP_TIMING
    g.dbg = nullptr;
#endif
    g.a = ClipPart{za, Pa, Ca, lda ? lda : Ca};
    g.b = ClipPart{zb, Pb, Cb, ldb ? ldb : Cb};
    g.tile_off = m.tile_off;
    g.tile_cnt = m.tile_cnt;
    g.tile_pool = reinterpret_cast<const int2*>(m.tile_pool);
    g.tile_rec = reinterpret_cast<const int4*>(m.tile_rec);
    g.tile_brec = reinterpret_cast<const int4*>(m.tile_brec);
    g.tile_bpool = reinterpret_cast<const int2*>(m.tile_bpool);
    g.tile_halo = m.tile_halo;
    g.brec_addr = m.brec_addr;
    g.xbuf = reinterpret_cast<unsigned long long*>(m.xbuf);
    g.sync = reinterpret_cast<unsigned*>(m.sync);
    g.err = reinterpret_cast<unsigned*>(m.err);
    g.T = m.T;
    g.nbj = m.nbj;
    // One launch holds whole (clip, slice) groups and at most one workgroup per CU (160 KB of LDS each): all of its workgroups
    // are resident together, so a workgroup that waits for a neighbour tile never waits for one that has not been dispatched.
    const int S = (Ca + Cb) / 4;
    const int per = qt_num_cus() / (m.B * m.T);    // slices per launch (>= 1: the entry points refuse B T > CUs)
    for (int s0 = 0; s0 < S; s0 += per) {
        g.s0 = s0;
        g.ns = S - s0 < per ? S - s0 : per;
        const int groups = m.B * g.ns;
        const int grid = ((groups + 7) / 8) * 8 * m.T;
        if (bwd)
            hipLaunchKernelGGL((k_cheb_clip<true, 4, true>), dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
        else
            hipLaunchKernelGGL((k_cheb_clip<false, 4, true>), dim3(grid), dim3(CL_T), 0, (hipStream_t)stream, g);
    }
    return 0;
}

#define TILE_MESH_ARGS_OK                                                                                                         \
    QT_ARG(rowptr && col && nrm && ell && tile_off && tile_cnt && tile_pool && tile_rec && tile_brec && tile_bpool && tile_halo && brec_addr && \
               xbuf && sync && err && B > 0 && K >= 2 && K <= 16 && T >= 2 && nbj >= 1 && T % nbj == 0 && ((uintptr_t)xbuf & 15) == 0,  \
           "bad arguments (the tile arrays of qt_edges_norm_tiles, the exchange buffer and the sync words zeroed by qt_edges_count, "     \
           "and the caller's persistent error word are required)");                                                                    \
    /* the tiles of a (clip, slice) wait for each other: all of them must be resident at once, one workgroup (160 KB of LDS) per CU */ \
    QT_ARG(B * T <= qt_num_cus(), "more tiles (B x T) than compute units: the tile-resident launch needs them co-resident -- "          \
                                  "use one qt_spmm launch per hop for this mesh")

// the capacities the tile structures are laid out for (the caller allocates them): 0 interior pool entries, 1 interior records,
// 2 halo slots = boundary records, 3 boundary pool entries (all per tile), 4 slices per launch
extern "C" int qt_tile_cap(int which) {
    switch (which) {
        case 0: return QT_TILE_POOL_CAP;
        case 1: return QT_TILE_REC_CAP;
        case 2: return QT_TILE_HALO_CAP;
        case 3: return QT_TILE_BPOOL_CAP;
        case 4: return QT_TILE_SLICES;
        default: return 0;
    }
}
extern "C" int qt_cheb_tile_sync_words(int B) { return 2 * B * QT_TILE_SLICES; }
extern "C" int qt_cheb_tile_xbuf_words(int B, int T) { return B * T * QT_TILE_SLICES * 2 * QT_TILE_HALO_CAP * 8; }   // (hop parity) x 4 granules of 2 ints per slot

extern "C" int qt_cheb_tile_fwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* tile_off, const int32_t* tile_cnt, const int32_t* tile_pool, const int32_t* tile_rec,
                                const int32_t* tile_brec, const int32_t* tile_bpool, const int32_t* tile_halo, const int32_t* brec_addr,
                                int32_t* xbuf, int32_t* sync, int32_t* err,
                                int B, int T, int nbj, int N, int K, int Ca, const float* za, int lda, float* Ta, int Cb,
                                const float* zb, int ldb, float* Tb, void* stream) {
    TILE_MESH_ARGS_OK;
    QT_ARG(za && Ta && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || (zb && Tb)), "bad operands");
    QT_ARG((Ca + Cb) / 4 <= QT_TILE_SLICES, "at most QT_TILE_SLICES 4-channel slices per launch");
    QT_ARG((lda | ldb) % 4 == 0, "row strides must be multiples of 4");
    QT_ARG((((uintptr_t)za | (uintptr_t)Ta | (uintptr_t)zb | (uintptr_t)Tb | (uintptr_t)ell | (uintptr_t)tile_pool | (uintptr_t)tile_rec |
             (uintptr_t)tile_brec | (uintptr_t)tile_bpool) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31) && (int64_t)N * max(max(lda, ldb), 4) < ((int64_t)1 << 31),
           "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    const TileMesh m = {rowptr, col, nrm, ell, tile_off, tile_cnt, tile_pool, tile_rec, tile_brec, tile_bpool, tile_halo, brec_addr, xbuf, sync, err, B, T, nbj};
    tile_launch(false, m, N, K, Ca, za, lda, Ta, Cb, zb, ldb, Tb, stream, 0);
    QT_LAUNCHED();
    return QT_OK;
}

extern "C" int qt_cheb_tile_bwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell,
                                const int32_t* tile_off, const int32_t* tile_cnt, const int32_t* tile_pool, const int32_t* tile_rec,
                                const int32_t* tile_brec, const int32_t* tile_bpool, const int32_t* tile_halo, const int32_t* brec_addr,
                                int32_t* xbuf, int32_t* sync, int32_t* err,
                                int B, int T, int nbj, int N, int K, int Ca, float* Ga, int Cb, float* Gb, int planes_sm,
                                void* stream) {
    TILE_MESH_ARGS_OK;
    QT_ARG(Ga && Ca > 0 && Ca % 4 == 0 && Cb >= 0 && Cb % 4 == 0 && (Cb == 0 || Gb), "bad operands");
    QT_ARG((Ca + Cb) / 4 <= QT_TILE_SLICES, "at most QT_TILE_SLICES 4-channel slices per launch");
    QT_ARG((((uintptr_t)Ga | (uintptr_t)Gb | (uintptr_t)ell | (uintptr_t)tile_pool | (uintptr_t)tile_rec | (uintptr_t)tile_brec |
             (uintptr_t)tile_bpool) & 15) == 0, "operands must be 16-byte aligned");
    QT_ARG((int64_t)K * N * max(Ca, Cb) < ((int64_t)1 << 31), "planes too large for 32-bit offsets");
    if (N <= 0) return QT_OK;
    const TileMesh m = {rowptr, col, nrm, ell, tile_off, tile_cnt, tile_pool, tile_rec, tile_brec, tile_bpool, tile_halo, brec_addr, xbuf, sync, err, B, T, nbj};
    tile_launch(true, m, N, K, Ca, nullptr, 0, Ga, Cb, nullptr, 0, Gb, stream, planes_sm);
    QT_LAUNCHED();
    return QT_OK;
}
