// Backward of the grouped projection of the attention-convolution stacks (qt_proj_group: P_g = [A_g | 1] W_g, model/model.py:51,
// 394-424 layer by layer) in ONE pass over the gradient planes:
//
//   gA_g (N, 32)      = gP_g (N, 128) W_g[:32]^T                      (data gradient, written)
//   gW_g (32 + 4, 128) += [A_g | 1 0 0 0]^T gP_g                      (weight gradient, accumulated per workgroup)
//
// Until round 5 these were two launches per layer and use -- the data gradient (qt_proj_group on the gradient planes) and, once per
// pass, the grouped weight gradient over the SAVED gradient planes of every rollout use (qt_wgrad_groups) -- which read the
// (8, 4, N, 32) gradient array (510 MB at N = 1.2e5) twice, the second time cold.  Both products are tiles of the same operand:
// a workgroup stages a 64-row x 128-column tile of gP_g in LDS once; two waves multiply its rows with W_g^T (reduction over the 128
// columns: the data gradient of those rows), the other two multiply its columns with the tile of A_g^T (reduction over the rows: a
// partial weight gradient that stays in the waves' accumulators over all the tiles the workgroup walks) -- 64 fp32 MFMAs
// (v_mfma_f32_32x32x2_f32) per wave and tile; the bias row is the column sum of the tile, accumulated on the vector pipe from the B
// operands the MFMAs load anyway.
// A cell's FIRST layer has the same shape turned sideways: the four stacks of a segment share ONE input (gsA = 0, rows lda apart:
// the state may be a column block of a wider matrix) and their weights sit side by side in one (36, 4 x 128) matrix (ldw = 512,
// gsW = 128).  Head h is group h: its workgroups write a PARTIAL data gradient (the caller adds the four) and slabs laid out like
// the weight matrix.  Measured at the cfg4t shape (tools/exp_proj_bwd.py): 108 - 115 us + 17 us for the sum against 101 - 116 (data
// gradient) + 99 - 112 us (share of the deferred weight gradient).
// Persistent: `nb` workgroups per group (all G nb of them resident), workgroup b of a group takes the row tiles b, b + nb, ...; the
// next tile's operands are requested before the current tile's MFMA chain.  Each workgroup
// owns one slab of the partial weight gradient (fixed summation order: deterministic); the uses of a pass add into the same slabs and
// the pass's last backward reduces them with qt_colsum.
#include "qt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float pb_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pb_gload4(const float* p) {
    const pb_v4f v = *(const __attribute__((address_space(1))) pb_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int PB_ROWS = 64;         // rows of a tile
constexpr int PB_C = 32;            // channels of a plane = input channels of the projection
constexpr int PB_CO = 4 * PB_C;     // gradient columns of a group: q | k | v | skip
constexpr int PB_GP = PB_CO + 4;    // LDS pitch of the gradient tile and of W: == 4 (mod 64) floats (conflict-free ds_read_b128 groups)
constexpr int PB_AP = PB_C + 4;     // LDS pitch of the A tile

struct ProjBwdArgs {
    const float* gP;        // group g: 4 planes (N, 32), plane stride psG, from gP + g gsG
    int64_t gsG, psG;
    const float* A;         // group g: (N, 32) rows, lda floats apart, from A + g gsA
    int64_t gsA;
    int lda;
    const float* W;         // group g: 32 + 4 rows of 128 columns, row pitch ldw, from W + g gsW -- (G, 36, 128) arrays: ldw = 128,
    int64_t gsW;            // gsW = 36 * 128; G heads side by side in ONE (36, G 128) matrix (a shared input: gsA = 0): ldw = G 128, gsW = 128
    int ldw;
    float* gA;              // group g: (N, 32) rows, row stride ldo, from gA + g gsO
    int64_t gsO;
    int ldo;
    float* part;            // nb slabs of G 36 128 floats, each laid out like W (group g at + g gsW, row pitch ldw)
    int N;
    const int32_t* n_dev;
    int G, nb, accumulate, reverse;
#ifdef QT_PB_TIMING
    long long* dbg;         // diagnostics build (tools/exp_proj_bwd.py): accumulated clock ticks per phase and workgroup
#endif
};
#ifdef QT_PB_TIMING
#define PB_T0() long long pb_t = wall_clock64(); long long pb_acc[6] = {0, 0, 0, 0, 0, 0}
#define PB_STAMP(i) do { const long long n_ = wall_clock64(); pb_acc[i] += n_ - pb_t; pb_t = n_; } while (0)
#define PB_DUMP() do { if (a.dbg && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 6; ++i_) a.dbg[((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 6 + i_] = pb_acc[i_]; } while (0)
#else
#define PB_T0() do {} while (0)
#define PB_STAMP(i) do {} while (0)
#define PB_DUMP() do {} while (0)
#endif

// 64-row tiles and 60 KB of LDS: TWO workgroups per CU, so that one's MFMA chain runs while the other stages its next tile (a first
// version with 128-row tiles, one workgroup per CU and every wave doing both products ran 171 us per layer-use at the cfg4t shape:
// the four waves sat at the same barriers).  Roles: waves 0 / 1 own the data gradient of rows [0, 32) / [32, 64) (64 MFMAs per tile:
// the full reduction over the 128 gradient columns, in qt_proj_group's order -> bit-identical rows); waves 2 / 3 own the weight
// gradient's column tiles {0, 1} / {2, 3} (2 x 32 MFMAs per tile, accumulators kept over all tiles of the workgroup).
__global__ __launch_bounds__(256, 2) void k_proj_bwd(ProjBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float Gs[PB_ROWS * PB_GP];
    __shared__ __attribute__((aligned(16))) float As[PB_ROWS * PB_AP];
    __shared__ __attribute__((aligned(16))) float Ws[PB_C * PB_GP];
    __shared__ __attribute__((aligned(16))) float Os[2 * 32 * PB_AP];       // the two data-gradient waves' output tiles, row-major
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, half = lane >> 5;
    int g = (int)blockIdx.x % a.G;
    if (a.reverse) g = a.G - 1 - g;
    const int b = (int)blockIdx.x / a.G;
    const int rows = qt_rows(a.n_dev, a.N);
    const int ntiles = (rows + PB_ROWS - 1) / PB_ROWS;
    const float* gP = a.gP + g * a.gsG;
    const float* A = a.A + g * a.gsA;
    const float* W = a.W + g * a.gsW;
    float* gA = a.gA + g * a.gsO;

    // W_g[:32] -> LDS once: Ws[j][k] = W[j][k] (row j = input channel, k = gradient column)
    for (int e = t; e < PB_C * (PB_CO / 4); e += 256) {
        const int j = e >> 5, q = e & 31;
        *reinterpret_cast<float4*>(&Ws[j * PB_GP + 4 * q]) = pb_gload4(W + (int64_t)j * a.ldw + 4 * q);
    }

    // the tile's operands: 8 float4 of the gradient planes and 2 float4 of A per thread.  A plane's 64 rows are one contiguous 8 KB
    // block (row stride = 32 floats): instruction u reads 4 KB of plane u / 2 with consecutive threads on consecutive addresses;
    // rows past the valid count load as zeros (they then add nothing to either product)
    // (two tiles in flight -- a second register set -- measured the same 220 us per layer-use at the cfg4t shape: the operands are
    // not what the waves wait for)
    float4 pg[8], pa[2];
    // (no load sits behind a branch: rows past the valid count are clamped to the last valid row when loading and zeroed when the
    // tile goes to LDS; `live` = which of this thread's two row slots of the tile in the registers are valid)
    bool live0 = false, live1 = false;
    const int64_t last = rows > 0 ? rows - 1 : 0;
    const int c8 = 4 * (t & 7), rs0 = t >> 3, rs1 = 32 + (t >> 3);        // this thread's column piece and its two row slots of a tile
    auto fetch = [&](int tile) {
        const int64_t row0 = (int64_t)tile * PB_ROWS + rs0, row1 = (int64_t)tile * PB_ROWS + rs1;
        live0 = tile < ntiles && row0 < rows;
        live1 = tile < ntiles && row1 < rows;
        const int64_t q0 = row0 < last ? row0 : last, q1 = row1 < last ? row1 : last;
        const int64_t o0 = q0 * PB_C + c8, o1 = q1 * PB_C + c8;
        pg[0] = pb_gload4(gP + o0);
        pg[1] = pb_gload4(gP + o1);
        pg[2] = pb_gload4(gP + a.psG + o0);
        pg[3] = pb_gload4(gP + a.psG + o1);
        pg[4] = pb_gload4(gP + 2 * a.psG + o0);
        pg[5] = pb_gload4(gP + 2 * a.psG + o1);
        pg[6] = pb_gload4(gP + 3 * a.psG + o0);
        pg[7] = pb_gload4(gP + 3 * a.psG + o1);
        pa[0] = pb_gload4(A + q0 * a.lda + c8);
        pa[1] = pb_gload4(A + q1 * a.lda + c8);
    };
    auto stash = [&]() {
        // (values, not `cond ? pg[i] : z` on the arrays themselves: that is a select of two ADDRESSES, which keeps the arrays in
        // scratch memory -- measured: 405 us per launch instead of 217)
        const float m0 = live0 ? 1.0f : 0.0f, m1 = live1 ? 1.0f : 0.0f;
        auto masked = [](float4 v, float m) { return m != 0.0f ? v : make_float4(0.f, 0.f, 0.f, 0.f); };
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            *reinterpret_cast<float4*>(&Gs[rs0 * PB_GP + pl * PB_C + c8]) = masked(pg[2 * pl], m0);
            *reinterpret_cast<float4*>(&Gs[rs1 * PB_GP + pl * PB_C + c8]) = masked(pg[2 * pl + 1], m1);
        }
        *reinterpret_cast<float4*>(&As[rs0 * PB_AP + c8]) = masked(pa[0], m0);
        *reinterpret_cast<float4*>(&As[rs1 * PB_AP + c8]) = masked(pa[1], m1);
    };

    const bool dwave = wave < 2;    // (wave-uniform role)
    f32x16 acc0, acc1;              // data gradient waves: acc0 = the tile's rows; weight gradient waves: their two column tiles
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
    float bs0 = 0.0f, bs1 = 0.0f;   // bias row: column sums over the rows of parity `half`
    const int ct = 2 * (wave & 1);  // first column tile of a weight gradient wave
    fetch(b);
    PB_T0();
    for (int tile = b; tile < ntiles; tile += a.nb) {
        __syncthreads();            // the previous tile's MFMA chains are through with the LDS tiles (and Ws is in place)
        PB_STAMP(0);
        stash();
        PB_STAMP(1);
        __syncthreads();
        PB_STAMP(2);
        fetch(tile + a.nb);         // in flight during the MFMA chains below
        PB_STAMP(3);
        if (dwave) {
            // reduction index k = 8 j + 4 half + i in MFMA i of step j (the operand maps of k_gemm_fwd: a lane's float4 is four
            // consecutive k of its own row): one accumulator chain in qt_proj_group's order
            const float* grow = &Gs[(wave * 32 + l32) * PB_GP + 4 * half];
            const float* wrow = &Ws[l32 * PB_GP + 4 * half];
            f32x16 accd;
#pragma unroll
            for (int r = 0; r < 16; ++r) accd[r] = 0.0f;
            // the LDS reads of step group n + 1 are issued BEFORE the MFMAs of group n (sched_barrier keeps the order): left to the
            // scheduler every group of MFMAs waited for reads issued just before it
            float4 ga[2][4], wb[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ga[0][u] = *reinterpret_cast<const float4*>(grow + 8 * u);
                wb[0][u] = *reinterpret_cast<const float4*>(wrow + 8 * u);
            }
#pragma unroll
            for (int jg = 0; jg < 4; ++jg) {
                const int cur = jg & 1;
                if (jg + 1 < 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        ga[cur ^ 1][u] = *reinterpret_cast<const float4*>(grow + 8 * (4 * (jg + 1) + u));
                        wb[cur ^ 1][u] = *reinterpret_cast<const float4*>(wrow + 8 * (4 * (jg + 1) + u));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[cur][u].x, wb[cur][u].x, accd, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[cur][u].y, wb[cur][u].y, accd, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[cur][u].z, wb[cur][u].z, accd, 0, 0, 0);
                    accd = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[cur][u].w, wb[cur][u].w, accd, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PB_STAMP(4);
            // register r of a lane = row (r & 3) + 8 (r >> 2) + 4 half, column l32.  Stored straight from the accumulators the tile
            // costs 16 dword stores per lane (1.4 us per tile of vector-memory issue); through the wave's own 32 x 32 LDS tile the
            // rows leave as 4 float4 stores per lane (lane = row l >> 1 + 32 u / 8 .., half a 128-byte row each)
            float* os = Os + wave * 32 * PB_AP;
#pragma unroll
            for (int r = 0; r < 16; ++r) os[((r & 3) + 8 * (r >> 2) + 4 * half) * PB_AP + l32] = accd[r];
            const int64_t r0 = (int64_t)tile * PB_ROWS + wave * 32;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = 8 * u + (lane >> 3), cc = 4 * (lane & 7);
                const float4 v = *reinterpret_cast<const float4*>(&os[rr * PB_AP + cc]);      // (same wave: LDS ops complete in order)
                if (r0 + rr < rows) *reinterpret_cast<float4*>(gA + (r0 + rr) * a.ldo + cc) = v;
            }
            PB_STAMP(5);
        } else {
            // rows 2 ks + half of the tile: A-operand = A[row][feature l32], B-operand = gP[row][column]
            const float* acol = &As[half * PB_AP + l32];
            const float* gcol = &Gs[half * PB_GP + ct * 32 + l32];
            constexpr int KG = 8;                   // k-steps per group; groups double-buffered in registers as above
            float av[2][KG], g0[2][KG], g1[2][KG];
#pragma unroll
            for (int u = 0; u < KG; ++u) {
                av[0][u] = acol[2 * u * PB_AP];
                g0[0][u] = gcol[2 * u * PB_GP];
                g1[0][u] = gcol[2 * u * PB_GP + 32];
            }
#pragma unroll
            for (int kg = 0; kg < PB_ROWS / 2 / KG; ++kg) {
                const int cur = kg & 1;
                if (kg + 1 < PB_ROWS / 2 / KG) {
#pragma unroll
                    for (int u = 0; u < KG; ++u) {
                        const int ks = KG * (kg + 1) + u;
                        av[cur ^ 1][u] = acol[2 * ks * PB_AP];
                        g0[cur ^ 1][u] = gcol[2 * ks * PB_GP];
                        g1[cur ^ 1][u] = gcol[2 * ks * PB_GP + 32];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < KG; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][u], g0[cur][u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][u], g1[cur][u], acc1, 0, 0, 0);
                    bs0 += g0[cur][u];
                    bs1 += g1[cur][u];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PB_STAMP(4);
        }
    }
    PB_DUMP();
    if (dwave) return;
    // this workgroup's slab of the weight gradient: rows 0 .. 31 from the accumulators, row 32 = the column sums (even + odd rows),
    // rows 33 .. 35 (the padding of the bias block) zero
    float* slab = a.part + (int64_t)b * a.G * (PB_C + 4) * PB_CO + g * a.gsW;
    const int ldw = a.ldw;
    // (all 34 slab reads of a lane first, then the stores: written element by element the compiler kept load -> add -> store in
    // order -- the pointers may alias -- and the 34 dependent round trips cost ~50 us per launch)
    float* s0 = slab + ct * 32 + l32;                              // this lane's column of tile ct; tile ct + 1 is 32 floats on
    f32x16 o0, o1;
    float ob0 = 0.0f, ob1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
        o0[r] = a.accumulate ? s0[f * ldw] : 0.0f;
        o1[r] = a.accumulate ? s0[f * ldw + 32] : 0.0f;
    }
    if (a.accumulate && half == 0) {
        ob0 = s0[PB_C * ldw];
        ob1 = s0[PB_C * ldw + 32];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
        s0[f * ldw] = o0[r] + acc0[r];
        s0[f * ldw + 32] = o1[r] + acc1[r];
    }
    const float b0 = bs0 + __shfl_xor(bs0, 32, 64), b1 = bs1 + __shfl_xor(bs1, 32, 64);
    if (half == 0) {
        s0[PB_C * ldw] = ob0 + b0;
        s0[PB_C * ldw + 32] = ob1 + b1;
    } else if (!a.accumulate) {
#pragma unroll
        for (int f = 1; f < 4; ++f) s0[(PB_C + f) * ldw] = s0[(PB_C + f) * ldw + 32] = 0.0f;
    }
}

}  // namespace

extern "C" int qt_num_cus(void);
#ifdef QT_PB_TIMING
static long long* g_pb_dbg = nullptr;
extern "C" void qt_proj_bwd_timing_buffer(long long* p) { g_pb_dbg = p; }
#endif

// Workgroups per group: the launch fills the chip once, every workgroup resident (two per CU: 60 KB of LDS each).
extern "C" int qt_proj_bwd_blocks(int G) {
    if (G <= 0) return 0;
    const int nb = 2 * qt_num_cus() / G;
    return nb < 1 ? 1 : nb;
}

extern "C" int qt_proj_bwd(const float* gP, int64_t gsG, int64_t psG, const float* A, int64_t gsA, int lda, const float* W, int64_t gsW,
                           int ldw, float* gA, int64_t gsO, int ldo, float* part, int N, const int32_t* n_dev, int G, int Cin, int C,
                           int accumulate, int reverse, void* stream) {
    QT_ARG(gP && A && W && gA && part && G >= 1, "null pointer");
    QT_ARG(Cin == PB_C && C == PB_C, "this launch is built for 32 input channels and 32-channel planes (hidden size 32)");
    QT_ARG(ldo >= Cin && ldo % 4 == 0 && gsO % 4 == 0 && ((uintptr_t)gA & 15) == 0 && gsG % 4 == 0 && psG % 4 == 0 && gsA % 4 == 0 && lda % 4 == 0 && lda >= Cin && gsW % 4 == 0 && ldw % 4 == 0, "strides must be multiples of 4 floats");
    QT_ARG(ldw >= PB_CO && (ldw == PB_CO || (int64_t)ldw == (int64_t)G * PB_CO), "ldw: 128 (one matrix per group) or G 128 (the groups side by side in one matrix)");
    QT_ARG((((uintptr_t)gP | (uintptr_t)A | (uintptr_t)W) & 15) == 0, "gP / A / W must be 16-byte aligned");
    if (N <= 0) return QT_OK;
    ProjBwdArgs a = {gP, gsG, psG, A, gsA, lda, W, gsW, ldw, gA, gsO, ldo, part, N, n_dev, G, qt_proj_bwd_blocks(G), accumulate, reverse};
#ifdef QT_PB_TIMING
    a.dbg = g_pb_dbg;
#endif
    hipLaunchKernelGGL(k_proj_bwd, dim3(G * a.nb), dim3(256), 0, (hipStream_t)stream, a);
    QT_LAUNCHED();
    return QT_OK;
}
