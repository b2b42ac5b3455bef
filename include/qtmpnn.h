/*
 * qtmpnn.h -- C ABI of libqtmpnn_hip.so: the MI355X (gfx950) kernels behind the
 * Quadtree-MPNNLSTM training hot path.
 *
 * The reference (zach-gousseau/Quadtree-MPNNLSTM) is pure Python and has no FFI;
 * each entry below names the reference function whose arithmetic it replaces
 * (file:line into the reference tree).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; the caller owns all
 *     memory, the library never allocates or frees;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     never synchronises and keeps no global state besides the last-error text;
 *   - return value 0 = ok, negative = QT_E_* ; qt_last_error() gives the text;
 *   - fp32 data, int32 indices; node-feature matrices are row-major (N, C);
 *   - every entry that takes a node count `int N` also takes `const int32_t* n_dev`: NULL means "N rows
 *     are valid"; otherwise N is the CAPACITY of the buffers (row strides, grid size) and the valid row
 *     count is read from *n_dev on the device (node_off + B of qt_quadtree_stage3), so a whole training
 *     step can be captured in a hipGraph with no host read-back of the data-dependent mesh size;
 *   - a "mesh" is the block-diagonal quadtree graph of B clips:
 *       labels (B, n, m) int32   global node id of every pixel, -1 = masked
 *       level  (B, n, m) uint8   log2(cell size) of the pixel's leaf
 *       cell   (N, 4)    int32   {row0, col0, size, clip} of every node
 *       rowptr (N+1), col (E), nrm (E)   CSR of L^ = -D^-1/2 W D^-1/2 (no self loops)
 */
#ifndef QTMPNN_H
#define QTMPNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QT_OK 0
#define QT_E_ARG (-1)     /* bad argument (shape, alignment, null pointer) */
#define QT_E_LAUNCH (-2)  /* HIP launch error */

#define QT_COND_MAX_LARGER 0
#define QT_COND_MAX_SMALLER 1
#define QT_COND_MIN_LARGER 2
#define QT_COND_MIN_SMALLER 3

/* tail edges (edges 5, 6, .. of a row) kept per clip for the clip-resident recurrence kernel: pool entries per clip */
#ifndef QT_TAIL_CAP /* (a test build shrinks it to force the pool-overflow path: csrc/Makefile, target smallcaps) */
#define QT_TAIL_CAP 4080
#endif
#define QT_TAIL_CNT_STRIDE 32 /* ints between the per-clip tail counters (one 128-byte line each): [0] tail edges, [1] rows with a tail */
#define QT_TAIL_REC_CAP 4096 /* records (rows with more than four edges) per clip = the most rows a clip-resident clip can have */

/* Frames of several 64 x 64 base cells ("tiles"): the tile-resident recurrence kernel (qt_cheb_tile_fwd / _bwd) keeps ONE TILE's
 * rows in LDS and exchanges the rows on tile borders between the workgroups of a clip after every hop.  qt_edges_norm_tiles
 * leaves per tile (slot = clip * T + tile in label order, as qt_quadtree_stage3's cell_off):
 *   counters  tile_cnt[QT_TILE_CNT_STRIDE * slot + ..]: [0] interior pool entries, [1] interior records (rows with more than four
 *             edges, all of them inside the tile), [2] halo entries (directed edges that leave the tile), [3] boundary records
 *             (rows with at least one such edge), [4] boundary pool entries, [5] != 0: a capacity below was exceeded (never for
 *             quadtree meshes: a tile has at most 256 pixel adjacencies across its border) */
#define QT_TILE_CNT_STRIDE 32
#define QT_TILE_REC_CAP 4096  /* interior records per tile */
#define QT_TILE_POOL_CAP 1372 /* interior pool entries per tile held in LDS (more: those rows walk the CSR arrays) */
#ifndef QT_TILE_HALO_CAP /* (the small-caps test build shrinks it to force the capacity-overflow report) */
#define QT_TILE_HALO_CAP 256  /* halo entries = boundary records at most, per tile: a power of two <= 256 */
#endif
#define QT_TILE_BPOOL_CAP 1024 /* boundary pool entries per tile (sum over border cells of 4 side - 4 < 1024) */
#define QT_TILE_SLICES 12     /* 4-channel slices per launch the exchange buffer and the sync words are laid out for (Ca + Cb <= 48) */

#define QT_ACT_NONE 0
#define QT_ACT_RELU 1
#define QT_ACT_TANH_RES 2 /* y = tanh(drop * acc) + res */
#define QT_ACT_RELU_BWD 3 /* y = acc where res > 0, else 0: the backward of QT_ACT_RELU applied to a data-gradient product (qt_dense2) */

const char* qt_last_error(void);
int qt_abi_version(void);

/* ---------------------------------------------------------------- mesh build
 * quadtree_decompose, model/graph_functions.py:145-259 (+ get_mapping :555-587).
 *
 * Stage 1: one workgroup per base cell.  Criterion value of padded pixel (r, c):
 *   src != NULL : src[b][min(r, src_rows-1)][min(c, src_cols-1)]          (edge padding, :190)
 *   src == NULL : nodeval[old_labels[b][r'][c']] (0 where the old label is -1), r' = min(r, n-1) ...
 *                 i.e. the un-flattened previous output (seq2seq.py:440) without materialising it.
 * Writes per-pixel local leaf ids / levels and one leaf count per base cell in
 * DFS order (cnt, length B * nbase).  Stage 2 (qt_scan_i32) turns cnt into
 * offsets; stage 3 writes labels, level, cell and counts[0] = N.
 * mask / hir: (n, m) uint8 or NULL, shared by all clips.
 * Requires ceil(n/max_size) <= ceil(m/max_size) padded sizes n_pad <= m_pad (the reference
 * raises IndexError otherwise, :222-229).
 */
int qt_quadtree_stage1(const float* src, int src_rows, int src_cols,
                       const float* nodeval, int nodeval_stride /* floats between node values */, const int32_t* old_labels,
                       int B, int n, int m, int max_size, float thresh, int condition,
                       const uint8_t* mask, const uint8_t* hir,
                       int32_t* local_id /* (B,n,m) */, uint8_t* level /* (B,n,m) */,
                       int32_t* cnt /* (B*nbase), or (B*nbase*4) with quads */,
                       int quads /* != 0 (max_size == 64 only): one workgroup per 32 x 32 quadrant of a base cell, leaf counts per
                       quadrant in DFS order; stage 3 must get the same flag */,
                       int32_t* fill_neg1 /* optional */, int fill_len /* fill_neg1[0 .. fill_len) = -1: the bwd_src array stage 3
                       completes (entries it does not write must read "no direct source row") */, void* stream);
int qt_quadtree_stage3(const int32_t* local_id, const int32_t* cnt_offsets /* exclusive scan of cnt, length B*nbase[*4]+1 */,
                       int B, int n, int m, int max_size,
                       int32_t* labels /* (B,n,m) */, const uint8_t* level, int32_t* cell /* (Nmax,4) */,
                       int32_t* node_off /* (B+1) */,
                       float size_norm, float* feat /* (Nmax,3) or NULL */, float* npix /* (Nmax) or NULL: the
                       qt_node_features outputs, written by the same pass */,
                       int raw_counts /* != 0: cnt_offsets is stage 1's cnt itself (<= 1024 counts), scanned by
                       every workgroup in LDS: no qt_scan_i32 launch in between (static capacities) */,
                       int quads /* as given to stage 1: four count slots per base cell */,
                       const int32_t* old_labels, const uint8_t* old_level /* the mesh this one was built from, or NULL */,
                       int32_t* fwd_src /* (Nmax) or NULL: per node of THIS mesh the old node under its pixel if the node
                       is a single pixel, else -1: qt_remesh's `direct` index for the transfer old -> this mesh */,
                       int32_t* bwd_src /* (Nmax_old) or NULL: the same for the transposed transfer this mesh -> old */,
                       int32_t* cell_off /* (B*nbase + 1) or NULL: first node of every base cell in label order, slot
                       b*nbase + (nbase - 1 - base): qt_remesh_clip's source ranges for frames of several base cells */,
                       void* stream);

/* exclusive scan: out[0]=0, out[i+1]=sum(in[0..i]); len+1 outputs.  tmp: (len/1024+2) int32. */
int qt_scan_i32(const int32_t* in, int32_t* out, int64_t len, int32_t* tmp, void* stream);
/* in-place exclusive scan of sums[0..nblk) by one workgroup; sums[nblk] = grand total */
int qt_scan_top(int32_t* sums, int nblk, void* stream);

/* static node features {col centroid / m, row centroid / n, npix / size_norm} (image_to_graph :657-668,
 * utils.py:37-45) and npix; feat (N, 3), npix (N). */
int qt_node_features(const int32_t* cell, int N, const int32_t* n_dev, int n, int m, float size_norm,
                     float* feat, float* npix, void* stream);

/* ---------------------------------------------------------------- adjacency
 * get_adj + dist, model/graph_functions.py:261-363, and the ChebConv normalisation that PyG
 * recomputes in every conv call (torch_geometric ChebConv.__norm__, restated in oracle/).
 * qt_edges_count: distinct neighbours per (node, side) -> cnt4, plus the total of every 1024-thread workgroup -> sums
 *   (nblk = qt_edges_blocks(N) entries; cnt4 holds nblk*1024 ints); qt_edges_fill adds up the totals before its workgroup
 *   itself.
 * qt_edges_fill: rowptr (N+1), col, w = centroid distance * resolution, and dis[i] = 1/sqrt(sum of row i's weights).
 * qt_edges_norm: nrm[e] = -dis_i w_e dis_j.
 */
int qt_edges_blocks(int N);
int qt_edges_count(const int32_t* labels, const int32_t* cell, int N, const int32_t* n_dev, int n, int m,
                   int32_t* cnt4, int32_t* sums /* nblk+1 */, int32_t* tail_cnt /* optional (B * QT_TAIL_CNT_STRIDE): clip c's counter
                   [QT_TAIL_CNT_STRIDE * c] is zeroed here for qt_edges_norm */, int B,
                   int32_t* zero_buf /* optional: zero_len ints set to 0 by the same launch (the tile counters and sync words
                   of qt_edges_norm_tiles / qt_cheb_tile_*) */, int zero_len, void* stream);
int qt_edges_fill(const int32_t* labels, const int32_t* cell, const int32_t* cnt4, const int32_t* sums /* as counted */,
                  int N, const int32_t* n_dev, int n, int m, float resolution,
                  int32_t* rowptr /* N+1 */, int32_t* col, float* w, float* dis /* N */, void* stream);
int qt_edges_norm(const int32_t* rowptr, const int32_t* col, const float* w, const float* dis, int N,
                  const int32_t* n_dev, float* nrm, int32_t* ell /* optional (N, 8): the first four edges of every row
                  again as [col x4 | nrm bits x4] (an unused slot = the row itself with weight 0; a complemented last column
                  flags more than four edges): qt_spmm2 then reaches its gathers without the row pointer */,
                  const int32_t* cell, const int32_t* node_off /* (B + 1) */, int32_t* tail_cnt /* (B * QT_TAIL_CNT_STRIDE), zeroed by qt_edges_count */,
                  int32_t* tail_pool /* (B, QT_TAIL_CAP, 2) */, int32_t* tail_info /* (N) */,
                  int32_t* tail_rec /* optional (B, QT_TAIL_REC_CAP, 8) */, void* stream);
/* tail_info != NULL (then the four arrays before it are required): the edges beyond the fourth of every row, per clip, as
 * {column - node_off[clip], weight bits} runs in tail_pool; tail_info[i] = run base | edge count << 16 (0: at most four edges;
 * base 0xffff: the clip's pool was full, the row stays on the CSR arrays).
 * tail_rec != NULL (needs tail_info and ell): one 32-byte record per row with more than four edges, clip c's records in the
 * order their rows' atomic adds on tail_cnt[QT_TAIL_CNT_STRIDE c + 1] arrive (at most QT_TAIL_REC_CAP are kept):
 *   {lc01, lc23, w0, w1}, {w2, w3, info, row - node_off[c]}: lc = the first four columns relative to the clip, as
 *   (column << 4) | (next column << 20); w = their weights' bits; info = the row's tail_info word.
 * qt_cheb_clip_fwd / _bwd read tail_cnt, tail_pool and tail_rec: thread j of a clip's workgroup takes record j. */
int qt_tail_cap(void);
/* qt_edges_norm for frames of several base cells (n x m > 4096, max_size 64): nrm and ell as qt_edges_norm, and per TILE
 * (tile_off = qt_quadtree_stage3's cell_off, (B * T + 1); T = tiles per clip, nbj = tiles per tile row; counters zeroed by
 * qt_edges_count's zero_buf) what qt_cheb_tile_fwd / _bwd run from:
 *   tile_rec   (B T, QT_TILE_REC_CAP, 8)  interior records, the format of tail_rec with columns relative to the TILE's first node
 *   tile_pool  (B T, QT_TILE_POOL_CAP, 2) {column - tile first node, weight bits} runs of those rows
 *   tile_brec  (B T, QT_TILE_HALO_CAP, 8) boundary records {lc01, lc23, w0, w1}, {w2, w3, info, row - tile first node}: a 16-bit
 *              column field is (local row << 4), or (halo slot << 4) | 1 for a neighbour in another tile; info = run base in
 *              tile_bpool | edge count << 16
 *   tile_bpool (B T, QT_TILE_BPOOL_CAP, 2) {local row, or 0x80000000 | halo slot; weight bits}: edges 5.. of the boundary rows
 *   tile_halo  (B T, QT_TILE_HALO_CAP)    global row of every halo slot
 *   brec_addr  (N)                        for a boundary row: tile slot * QT_TILE_HALO_CAP + its boundary record's index (where the
 *              other tiles look for the row's published values); -1 when its record did not fit; other rows: not written
 *   err        optional: the caller's PERSISTENT error word (see qt_cheb_tile_fwd); bit 1 is OR-ed in when a boundary row's
 *              record, halo slots or pool run exceed the tile capacities (no quadtree mesh does; any other CSR may) */
int qt_edges_norm_tiles(const int32_t* rowptr, const int32_t* col, const float* w, const float* dis, int N,
                        const int32_t* n_dev, float* nrm, int32_t* ell, const int32_t* cell, const int32_t* tile_off,
                        int T, int nbj, int32_t* tile_cnt, int32_t* tile_pool, int32_t* tile_rec, int32_t* tile_brec,
                        int32_t* tile_bpool, int32_t* tile_halo, int32_t* brec_addr, int32_t* err, void* stream);

/* ---------------------------------------------------------------- mesh <-> image transfers
 * flatten / unflatten, model/graph_functions.py:391-419, 451-458, by labels instead of the dense
 * (N, P) mapping.
 *
 * qt_gather: img[b,p,:] = scale_p * val[labels[b,p], :]  (0 where label < 0);
 *            scale_p = 1/npix[label] if inv_npix != NULL (the flatten backward) else 1.
 * qt_pool:   out[node, coff:coff+C] = sum over the node's pixels of value(p) (* 1/npix[node] if mean)
 *            value(p) = img[b, s, p, c] (src_labels == NULL; img laid out (B, S, n*m, C), out row = s*N + node)
 *                     = src_val[src_labels[b,p], c] * (src_inv ? 1/src_npix[..] : 1)     (remesh transfer,
 *                       seq2seq.py:440-442 + :474-477 fused; its backward swaps the roles of the meshes).
 * cell == NULL: one workgroup per 64x64 tile and clip does every node.  cell (N, 4) = (row, col, size, clip) given:
 * nodes of up to 4x4 pixels are reduced node by node (thread = node x float4 chunk, whole-row gathers and stores) and
 * the tile kernel only handles nodes of 8x8 pixels and more.  Deterministic either way (fixed order, no atomics).
 */
int qt_gather(const float* val, int C, const int32_t* labels, const float* inv_npix,
              int64_t npixels_total, float* img, void* stream);
int qt_pool(const float* img, int S, int64_t img_clip_stride /* floats between clips, 0 = dense */, const float* src_val, const int32_t* src_labels, const float* src_npix, int src_inv,
            int C, const int32_t* labels, const uint8_t* level, const float* npix, int mean,
            int B, int n, int m, int N, const int32_t* cell /* or NULL */, const int32_t* n_dev,
            float* out, int out_stride, int out_coff, void* stream);

/* qt_pool's mesh -> mesh transfer with the source node values given as up to 8 matrices side by side (host arrays of nparts
 * device pointers, widths and row strides, all multiples of 4): the state [out | H_0 .. | C_0 ..] is transferred across a
 * re-mesh (model/seq2seq.py:440-442, 474-477) without being concatenated first.  The result is written as nout dense
 * matrices (N, out_widths[i]) side by side (same total width): every consumer of a state part then reads dense rows. */
int qt_remesh(const float* const* src_parts, const int* widths, const int* lds, int nparts,
              const int32_t* src_labels, const float* src_npix, int src_inv, const int32_t* labels, const uint8_t* level,
              const float* npix, int mean, int B, int n, int m, int N, const int32_t* cell, const int32_t* n_dev,
              float* const* out_parts, const int* out_widths, int nout,
              const int32_t* direct /* (N) or NULL: fwd_src / bwd_src of qt_quadtree_stage3 for this pair of meshes: the
              single-pixel nodes then read their source row through it (same rows, two dependent loads instead of three) */,
              void* stream);

/* qt_remesh tile-resident (csrc/remeshclip.hip): one workgroup per (clip, 64 x 64 tile, 4-channel slice)
 * stages the tile's source rows in LDS (a base cell's nodes are one contiguous label range; frames of several tiles need
 * both meshes decomposed with max_size 64), gathers the pixel values from there and sums them up a 64 x 64 pyramid; every
 * destination node is written once by the thread that owns its head pixel.  Same arguments as qt_remesh minus the per-node
 * records (cell, N, n_dev, direct), plus the source mesh's node offsets (B + 1; qt_quadtree_stage3 writes them). */
int qt_remesh_clip_rows(void);
int qt_remesh_clip(const float* const* src_parts, const int* widths, const int* lds, int nparts,
                   const int32_t* src_labels, const float* src_npix, int src_inv,
                   const int32_t* src_cell_off /* (B*tiles + 1): first source node of every 64 x 64 tile in label order
                   (qt_quadtree_stage3's cell_off; for one-tile frames = node_off) */,
                   const int32_t* labels, const uint8_t* level, const float* npix, int mean, int B, int n, int m,
                   float* const* out_parts, const int* out_widths, int nout,
                   const float* posfeat /* optional (N, 3) of the destination mesh: float4 chunk 0 of the result is written as
                   (value.x, posfeat[node]) -- the decoder's next input [value | position, size], model/seq2seq.py:484-487 */,
                   int src_first_only /* != 0: only column 0 of the source's float4 chunk 0 counts (the transposed transfer of
                   that assembly: its gradient) */, void* stream);

/* qt_pool from an image, tile-resident: one workgroup per (clip, 64 x 64 tile, frame, channel) sums the
 * tile up a pyramid in LDS (csrc/remeshclip.hip).  Arguments as qt_pool's image form. */
int qt_pool_clip(const float* img, int S, int64_t img_clip_stride, int C, const int32_t* labels, const uint8_t* level,
                 const float* npix, int mean, int B, int n, int m, int N, float* out, int out_stride, int out_coff, void* stream);

/* masked MSE, model/mpnnlstm.py:243-246: partial[b*ntile + tile] = sum over the tile's unmasked pixels of
 * (out[labels[p]] - y[p])^2 ; y (B, n*m).  Pixels with label < 0 are the masked ones. */
int qt_sse(const float* out, int out_stride, const int32_t* labels, const float* y, int64_t y_clip_stride,
           int B, int n, int m, float* partial /* B*ceil(P/1024) */, void* stream);

/* The same for all output steps of a rollout at once (up to 16 per call; host arrays of nseg device pointers / sizes, one
 * mesh per step): partial (nseg, B*ceil(P/1024)) and, for the backward, sys[z] (N_z) = per-node sum of the target over
 * the node's pixels.  y: step z of clip b starts at y + z*y_step_stride + b*y_clip_stride.  qt_sse_rollout_bwd writes every
 * step's gradient rows (as qt_sse_bwd). */
int qt_sse_rollout(int nseg, const float* const* outs, const int* out_strides, const int32_t* const* labels,
                   const uint8_t* const* levels, const int* Ns, float* const* sys, const float* y, int64_t y_clip_stride,
                   int64_t y_step_stride, int B, int n, int m, float* partial, void* stream);
int qt_sse_rollout_bwd(int nseg, const float* const* outs, const int* out_strides, const float* const* npixs,
                       const float* const* sys, const int* Ns, const int32_t* const* n_devs, const float* g, int W,
                       float* const* gouts, void* stream);

/* gradient of the qt_sse partial sums with respect to the node values, written as full rows of width W (column 0 carries
 * the value, the rest zeros): gout[i, 0] = 2 * g[0] * (npix[i] * out[i * out_stride] - sy[i]), sy = per-node sum of y. */
int qt_sse_bwd(const float* out, int out_stride, const float* npix, const float* sy, const float* g, int N,
               const int32_t* n_dev, int W, float* gout, void* stream);

/* ---------------------------------------------------------------- message passing (ChebConv)
 * qt_spmm (the message-aggregate kernel; PyG MessagePassing.propagate of ChebConv, model/model.py:53,96):
 *   out[i,:] = alpha * sum_e nrm[e] * x[col[e],:] + beta * p[i,:] + gamma * q[i,:]      (p, q may be NULL)
 * x, out, p, q: (N, C) contiguous planes; out must not alias x (it may alias p or q).
 */
int qt_spmm(const int32_t* rowptr, const int32_t* col, const float* nrm, int N, const int32_t* n_dev, int C,
            const float* x, float alpha, const float* p, float beta, const float* q, float gamma,
            float* out, void* stream);

/* qt_spmm for rows stored as two matrices side by side, [a (N, Ca) | b (N, Cb)], in ONE launch (Cb == 0: only part a):
 * the recurrent cells propagate Z = [X | H] as X and H, never concatenated.  Widths multiples of 4; p / q given for
 * both parts or for neither.  x / p / q may be column views of wider matrices (row strides). */
int qt_spmm2(const int32_t* rowptr, const int32_t* col, const float* nrm, int N, const int32_t* n_dev,
             int Ca, const float* xa, int ldxa, const float* pa, int ldpa, const float* qa, int ldqa, float* outa,
             int Cb, const float* xb, int ldxb, const float* pb, int ldpb, const float* qb, int ldqb, float* outb,
             float alpha, float beta, float gamma, const int32_t* ell /* optional, from qt_edges_norm */,
             void* stream);   /* ld*: row strides in floats, 0 = dense; out rows are dense */

/* Clip-resident Chebyshev recurrences (csrc/chebclip.hip): ALL K - 1 message-aggregate hops of one ChebConv pass in ONE launch
 * for block-diagonal meshes whose clips hold at most qt_cheb_clip_rows() nodes each (4096: a 64 x 64 frame).  One workgroup per
 * (clip, 4-channel column slice) keeps two slice planes in LDS; gathers are LDS reads, hops are separated by a workgroup barrier.
 * Replaces K - 1 qt_spmm2 calls of PyG ChebConv.forward's recurrence (model/model.py:53,96); bit-identical planes.
 *   node_off (B + 1): first node of each clip (device; qt_quadtree_stage3 writes it); ell, tail_cnt / tail_pool / tail_rec:
 *   required, from qt_edges_norm (the first four edges of a row in registers, the rest from the clip's pool copied to LDS;
 *   a row with more than four edges is finished by the thread that holds its record, not by the row's owner).
 *   N: plane stride in rows (the capacity in static mode; the valid rows come from node_off).
 * qt_cheb_clip_fwd: T_k = 2 L^ T_{k-1} - T_{k-2} (T_0 = Z = [za | zb], T_1 = L^ Z) -> Ta, Tb: K - 1 planes each, stored
 *   SLICE-major -- plane k as (C / 4, N, 4): a workgroup owns one 4-channel slice, so consecutive rows of its slice are
 *   contiguous and its stores coalesce (row-major (N, C) planes cost a forward launch 7 - 9 % more).  The consumers take the
 *   layout through their planes_sm flag (qt_dense2, qt_dense_lstm, qt_wgrad, qt_wgrad_group).
 * qt_cheb_clip_bwd: Clenshaw on the gradient planes Ga (K, N, Ca), Gb (K, N, Cb): plane 0 is overwritten with
 *   A_0 + L^ b_1 - b_2, b_k = A_k + 2 L^ b_{k+1} - b_{k+2}; planes 1 .. K - 1 are left as given (the b_k stay in LDS). */
int qt_cheb_clip_rows(void);
/* width: channels per workgroup of the two launches below: 0 = automatic -- 2 when B * (Ca + Cb) / 2 workgroups fit the CUs in
 * one round (a hop is bound by the CU's LDS, so half-width slices on twice the CUs are faster), else 4; 2 or 4 pins it
 * (diagnostics, parity tests of both widths: the planes are the same bit for bit).  An argument, not a library switch: the
 * library keeps no mutable global state besides the thread-local error string. */
int qt_cheb_clip_fwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell, const int32_t* node_off,
                     const int32_t* tail_cnt, const int32_t* tail_pool, const int32_t* tail_rec,
                     int B, int N, int K, int Ca, const float* za, int lda, float* Ta,
                     int Cb, const float* zb, int ldb, float* Tb, int width, void* stream);
int qt_cheb_clip_bwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell, const int32_t* node_off,
                     const int32_t* tail_cnt, const int32_t* tail_pool, const int32_t* tail_rec,
                     int B, int N, int K, int Ca, float* Ga, int Cb, float* Gb,
                     int planes_sm /* != 0: planes 1 .. K-1 of Ga / Gb are slice-major (written so by qt_lstm_bwd_dgrad / qt_dense2) */,
                     int width, void* stream);

/* The same recurrences on frames of SEVERAL 64 x 64 base cells (128 x 128, 256 x 256, 96 x 128 ..; BASELINE configs[2]-[4]):
 * one workgroup per (clip, TILE, 4-channel slice) keeps the tile's <= 4096 rows in LDS; rows on a tile border have neighbours
 * in other tiles of the clip, so after every hop the tiles of a (clip, slice) exchange their border rows through global memory
 * as 8-byte {value, tag} granules (write-through stores by the row's owner, sc1 polls by the tiles that need it; the tag counts
 * launches and hops, so nothing is ever reset: csrc/chebclip.hip, TILE = true) instead of ending the launch.  Replaces K - 1
 * qt_spmm2 calls per ChebConv pass (model/model.py:53,96); the planes are bit-identical.
 *   tile_off = qt_quadtree_stage3's cell_off; tile_cnt .. tile_halo, brec_addr: from qt_edges_norm_tiles; xbuf: per (tile, slice,
 *   hop parity) QT_TILE_HALO_CAP slots of four granules;
 *   xbuf: qt_cheb_tile_xbuf_words(B, T) ints and sync: qt_cheb_tile_sync_words(B) ints, both zeroed ONCE per mesh (qt_edges_count's
 *   zero_buf).  err: ONE int32 the caller owns for as long as it wants to know (allocated once, zeroed once, never touched by a
 *   mesh build: it outlives the meshes and hipGraph replays); the launches OR into it: bit 0 = a wait for a neighbour tile
 *   timed out (the launch then finishes with garbage instead of hanging), bit 1 = a tile capacity of the mesh build was
 *   exceeded (the rows concerned are not computed).  A non-zero word means the planes of some launch since the word was last
 *   zero are WRONG: the caller must read it before trusting results (the Python host does so once per training epoch, every
 *   64 graph replays, after predict() and after every eager training step that issued such a launch).
 *   T tiles per clip in rows of nbj; N = plane stride in rows; Ca + Cb <= 4 QT_TILE_SLICES, K <= 16.  The launches are cut so
 *   that every workgroup of one is resident (<= one per CU); B * T > qt_num_cus() is refused (QT_E_ARG).  Planes as
 *   qt_cheb_clip_fwd / _bwd.
 *   Co-residency is what makes the waits safe: a GPU SHARED with another process that issues the same launches can leave two
 *   half-resident grids waiting for each other until the bounded spins give up (error word bit 0, garbage planes) -- one
 *   process per GPU, as everywhere in this library's multi-GPU use. */
int qt_cheb_tile_xbuf_words(int B, int T);
int qt_tile_cap(int which); /* the QT_TILE_* capacities: 0 pool, 1 records, 2 halo slots / boundary records, 3 boundary pool, 4 slices */
int qt_cheb_tile_sync_words(int B);
int qt_cheb_tile_fwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell, const int32_t* tile_off,
                     const int32_t* tile_cnt, const int32_t* tile_pool, const int32_t* tile_rec, const int32_t* tile_brec,
                     const int32_t* tile_bpool, const int32_t* tile_halo, const int32_t* brec_addr, int32_t* xbuf, int32_t* sync,
                     int32_t* err, int B, int T, int nbj, int N, int K,
                     int Ca, const float* za, int lda, float* Ta, int Cb, const float* zb, int ldb, float* Tb, void* stream);
int qt_cheb_tile_bwd(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell, const int32_t* tile_off,
                     const int32_t* tile_cnt, const int32_t* tile_pool, const int32_t* tile_rec, const int32_t* tile_brec,
                     const int32_t* tile_bpool, const int32_t* tile_halo, const int32_t* brec_addr, int32_t* xbuf, int32_t* sync,
                     int32_t* err, int B, int T, int nbj, int N, int K,
                     int Ca, float* Ga, int Cb, float* Gb, int planes_sm, void* stream);

/* The decoder head's backward products in ONE launch (model/seq2seq.py:115-121,164-171 backwards): G = relu'(Y) (.) (gU @ Wb2)
 * (N, 16) -- stored: the weight gradient of fc_out1 reads it -- and the data gradient of fc_out1, planes = G @ Wb1^T, as K planes
 * in two column parts (N, Cb) | (N, Cbb), planes 1.. slice-major on request.  gU (N, 4): gradient at the coefficient columns of
 * fc_out2; Wb2 (4, 16); Y (N, 16): fc_out1's ReLU output; Wb1 (K (Cb + Cbb), 16): the rows of fc_out1's weight.  Bit-identical to
 * the two qt_dense2 launches it replaces. */
int qt_head_dgrad(const float* gU, const float* Wb2, const float* Y, const float* Wb1, int K, int Cb, int Cbb, int N,
                  const int32_t* n_dev, float* G, float* out, float* outb, int out_sm, void* stream);

/* qt_dense: out planes = act( [A planes | S] @ [W ; Ws] ), the gate GEMM.
 *   A: Ka planes, plane k at a0 (k == 0) or a_rest + (k-1)*N*Ca, each (N, Ca)   (T_0 = Z stays in the caller's tensor)
 *   W: (Ka*Ca, Kb*Cb) row-major;  S: (N, Ks) or NULL with Ws (Ks, Kb*Cb)
 *   out: Kb planes of (N, Cb) at out + j*N*Cb
 *   act: QT_ACT_* applied to the result (only with Kb == 1); res (N) / drop (N) for QT_ACT_TANH_RES, drop may be NULL.
 */
int qt_dense(const float* a0, const float* a_rest, int Ka, int Ca, const float* W,
             const float* S, int Ks, const float* Ws, int Kb, int Cb, int N, const int32_t* n_dev,
             int act, const float* res, int res_stride, const float* drop, float* out, void* stream);

/* qt_dense2: qt_dense whose input planes and output planes may each be two matrices side by side:
 *   input plane k = [a (N, Ca) | b (N, Cab)] (a0 / a0b for k = 0, a_rest / a_restb (Ka-1, N, .) for the rest; Cab == 0: none);
 *   output plane j = [out (N, Cb) | outb (N, Cbb)] (Cbb == 0: none).  W rows follow the logical order (k, [a | b]),
 *   W columns the logical order (j, [out | outb]).  lda0 / lda0b: row strides of plane 0 in floats (0 = dense), so that
 *   plane 0 may be a column view of a wider matrix.  With WT the weight chunk is staged in LDS by straight 16-byte copies
 *   (the data gradient passes the forward weight itself here: (W^T)^T).  qt_dense is the Cab = Cbb = 0, dense case. */
int qt_dense2(const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb, int Ka, int Ca, int Cab,
              const float* W, const float* WT /* optional: [W ; Ws]^T, (Kb*(Cb+Cbb), K) row-major; W may then be NULL */,
              const float* S, int Ks, const float* Ws, int Kb, int Cb, int Cbb, int N,
              const int32_t* n_dev, int act, const float* res, int res_stride, const float* drop, float* out, float* outb,
              int planes_sm /* bit 0: the INPUT planes 1 .. Ka-1 (a_rest / a_restb) are stored slice-major, (plane, 4-channel
              slice, N, 4), as qt_cheb_clip_fwd writes them; bit 1: the OUTPUT planes 1 .. Kb-1 are written slice-major, as
              qt_cheb_clip_bwd reads them; plane 0 is row-major either way */,
              const float* post_W, float* post_out /* optional pair, only with 16 output columns and W: a second product in the
              epilogue, post_out (N, 4) = [act(out) | 1 0 0 0] @ post_W (20, 4) -- the decoder head's fc_out1 -> coefficient
              columns of fc_out2 in one launch; same bits as a second qt_dense2 call on the stored rows */,
              void* stream);

/* Data-gradient product as a split-bf16 GEMM (gradients only): out planes (Kb, N, Cb) [| outb (Kb, N, Cbb)] = A (N rows of K floats,
 * row stride lda) @ B, B^T given as the two bf16 terms Whi + Wlo (Kb (Cb + Cbb), K) of qt_split_bf16 -- for the data gradient of
 * Y = [T_0 .. | S] W that is the forward weight's own rows.  A is split on the fly; three bf16 MFMAs per product group, relative
 * error ~2^-16.  K % 16 == 0.  Replaces qt_dense2 in the backward of model/model.py:394-424 where the fp32-MFMA rate bounds it
 * (hidden 32: K = 128). */
int qt_dense_sb(const float* A, int lda, int K, const void* Whi, const void* Wlo, int Kb, int Cb, int Cbb, int N, const int32_t* n_dev,
                float* out, float* outb, void* stream);

/* qt_wgrad: partial sums of [A planes | S]^T @ G over row blocks, then qt_colsum over the blocks.
 *   G (N, Co); part (nblk, Ka*Ca + Ks, Co) with nblk = qt_wgrad_blocks(N).  accumulate != 0 adds into part
 *   (each block owns its slab, so the sum over several uses of one weight keeps a fixed order).  */
int qt_wgrad_blocks(int N);
int qt_wgrad(const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb, int Ka, int Ca, int Cab,
             const float* S, int Ks, const float* G, int Co, int N, const int32_t* n_dev, int accumulate, float* part,
             int planes_sm, void* stream);      /* planes in two parts and planes_sm as in qt_dense2 (Cab == 0: one part) */
/* qt_wgrad for up to 16 uses of one weight (the rollout steps of a pass) in ONE launch: host arrays of nseg device
 * pointers / capacities, shared (Ka, Ca, Ks, Co).  part: (qt_wgrad_group_blocks(nseg, N), Ka*Ca + Ks, Co), overwritten. */
int qt_wgrad_group_blocks(int nseg, const int* N);
int qt_wgrad_group(int nseg, const float* const* a0, const int* lda0, const float* const* a_rest, const float* const* a0b,
                   const int* lda0b, const float* const* a_restb, const float* const* S, const float* const* G, const int* N,
                   const int32_t* const* n_dev, int Ka, int Ca, int Cab, int Ks, int Co, float* part, int planes_sm, void* stream);
/* qt_wgrad_group for the Gn weights of qt_proj_group at once: use s multiplies [A_g | S]^T (A_g = a0[s] + g gsA, Cin columns, row
 * stride lda0[s]) with the gradient rows G[s] + g gsG (Co columns, row stride ldg; gpl > 0: stored as Co / gpl planes (N[s], gpl),
 * row stride ldg); per_node != 0: gsA / gsG count floats per node of the use (group stride = gs x N[s]: head-major arrays of uses with
 * different capacities).  part: (qt_wgrad_group_blocks(nseg, N), Gn, Cin + Ks, Co), overwritten; qt_colsum over the blocks gives the
 * (Gn, Cin + Ks, Co) weight gradient. */
int qt_wgrad_groups(int nseg, const float* const* a0, const int* lda0, const float* const* S, const float* const* G, const int* N,
                    const int32_t* const* n_dev, int Cin, int Ks, int Co, int ldg, int gpl, int Gn, int64_t gsA, int64_t gsG,
                    int per_node, float* part, void* stream);
/* G independent products in one launch: group g multiplies [A_g | S] with W_g.  A_g = Ka planes (N, Ca) from A + g gsA (one plane
 * may have any row stride lda; several are dense and contiguous), S (N, 4) = [1 0 0 0] rows or NULL (then no bias rows), W_g = W + g gsW
 * ((Ka Ca + 4 or Ka Ca), Kb Cb) row-major -- or its transpose WT + g gsW with rows of Ka Ca (+ 4) floats; the result leaves as Kb planes
 * (N, Cb) from out + g gsO, rows ldo apart (Kb = 1: a column block of a wider matrix).  The eight GraphConv stacks of a GConvLSTM built
 * from attention convolutions (model/model.py:394-424, TransformerConv :51) run layer by layer: group g is stack g's projection
 * [q | k | v | skip].  Same arithmetic as G qt_dense2 calls. */
int qt_proj_group(const float* A, int lda, int64_t gsA, int Ka, int Ca, const float* S, const float* W, const float* WT, int64_t gsW,
                  int G, int Kb, int Cb, float* out, int ldo, int64_t gsO, int reverse /* groups from the last to the first */, int N,
                  const int32_t* n_dev, void* stream);
/* Backward of qt_proj_group for attention stacks of hidden size 32 (what ice_exp.py:153-162 runs) in ONE pass over the gradient
 * planes.  Group g: input rows A_g (N, 32), lda floats apart, from A + g gsA, weights W_g = 32 + 4 rows of 128 columns from W + g gsW with row pitch ldw,
 * gradient planes gP_g = 4 planes (N, 32).  gA_g = gP_g W_g[:32]^T is written (rows ldo apart, group stride gsO) and the partial
 * weight gradient [A_g | 1 0 0 0]^T gP_g of workgroup b is written to (accumulate == 0) or added into (!= 0) slab b of part
 * (qt_proj_bwd_blocks(G) slabs of G 36 128 floats, each laid out like W); qt_colsum over the slabs gives the weight gradient of
 * all uses that added.  Two layouts: deeper layers -- every stack has its own input and its own (36, 128) matrix: ldw = 128,
 * gsW = 36 128; a cell's first layer -- G heads share ONE input (gsA = 0) and sit side by side in one (36, G 128) matrix:
 * ldw = G 128, gsW = 128, and the caller adds the G partial data gradients.  Replaces a qt_proj_group call on the gradient planes
 * plus the use's share of the deferred qt_wgrad_groups launch, which read the planes twice.  reverse != 0: groups are dealt to the
 * workgroups last to first. */
int qt_proj_bwd_blocks(int G);
int qt_proj_bwd(const float* gP, int64_t gsG, int64_t psG, const float* A, int64_t gsA, int lda, const float* W, int64_t gsW,
                int ldw, float* gA, int64_t gsO, int ldo, float* part, int N, const int32_t* n_dev, int G, int Cin, int C, int accumulate,
                int reverse, void* stream);
/* out[j] = sum_i part[i*len + j], i < nblk */
int qt_colsum(const float* part, int nblk, int64_t len, float* out, void* stream);

/* ---------------------------------------------------------------- GConvLSTM cell + LayerNorm
 * GConvLSTM gate arithmetic model/model.py:394-428 with the encoder/decoder LayerNorms of
 * model/seq2seq.py:64-75,140-151 fused (eps 1e-5).
 *   G (N, 4h) gate pre-activations in the order i, f, c, o (conv_x + conv_h sums);
 *   Cprev (N, h) or NULL (zeros); wc (3, h) peepholes i, f, o; b (4, h);
 *   ln (4, h): gamma_h, beta_h, gamma_c, beta_c, or NULL for the bare cell (Hn = H', Cn = C').
 * Outputs: O (N, h) raw output gate, Hn / Cn (N, h) LayerNorm'ed states,
 *   gates (N, 4h) activated I, F, T, O saved for the backward (which recomputes C' before its LayerNorm as
 *   fma(F, Cprev, I T), the forward's own rounding, instead of reading a saved copy).
 */
int qt_lstm_fwd(const float* G, const float* G2 /* optional second addend of the pre-activations (conv_x + conv_h) */,
                int ld_g /* row stride of G and G2, 0 = 4h */, const float* Cprev, int ld_c /* row stride of Cprev, floats */,
                const float* wc, const float* b, const float* ln,
                int N, const int32_t* n_dev, int h, float* O, float* Hn, float* Cn, float* gates,
                void* stream);
/* qt_dense (act none, Kb = 1, Cb = 4h) with qt_lstm_fwd as its epilogue, for hidden sizes 8, 16, 32: the gate pre-activations stay in
 * LDS.  Same results as the two calls (same arithmetic in the same order).  Planes in two parts as in qt_dense2; O may be
 * NULL (the raw output gate is also gates[:, 3h:4h]). */
int qt_dense_lstm(const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb, int Ka, int Ca, int Cab,
                  const float* W, const float* WT /* optional transpose, as in qt_dense2 */, const float* S, int Ks,
                  const float* Ws, int h, int N, const int32_t* n_dev, const float* Cprev, int ld_c,
                  const float* wc, const float* b, const float* ln, float* O, float* Hn, float* Cn,
                  float* gates, int planes_sm /* as in qt_dense2 */, void* stream);
/* gO, gHn, gCn may each be NULL (that output was not used: zero gradient).  part: (nblk, 11*h) partial sums [g_wc(3h) | g_b(4h) | g_ln(4h)], nblk = qt_lstm_bwd_blocks(N, h) */
int qt_lstm_bwd_blocks(int N, int h);
int qt_lstm_bwd(const float* gO, int ld_go, const float* gHn, int ld_gh, const float* gCn, int ld_gc,   /* row strides */
                const float* gates, const float* Cprev, int ld_c, const float* wc, const float* ln,
                int N, const int32_t* n_dev, int h, float* gG, float* gCprev, float* part, int accumulate,
                void* stream);

/* qt_lstm_bwd with the data gradient of the gate GEMM as its second half (hidden 8 / 16): a workgroup computes the gG rows of
 * its 128 nodes, writes them (the weight gradient reads them later) and multiplies them from LDS with W^T:
 * out planes (Kb, N, Cb) [+ outb (Kb, N, Cbb)] = gG (N, 4h) @ Wrows^T, Wrows = the first Kb (Cb + Cbb) rows of the forward
 * weight (row k (Cb + Cbb) + c, 4h columns), Kb (Cb + Cbb) <= 128.  Same planes, bit for bit, as qt_lstm_bwd + qt_dense2.
 * part: (qt_lstm_dgrad_blocks(N), 11 h) partial sums as in qt_lstm_bwd (one row per 128-node workgroup).
 * Whi / Wlo (optional pair, bf16 (Kb (Cb + Cbb), 4h) from qt_split_bf16(Wrows)): the data gradient then runs as a split-bf16
 * product on bf16 MFMA (gG and W as two bf16 terms each, three products: relative error ~2^-16, gradients only) instead of
 * fp32 MFMA; NULL: exact fp32, bit for bit qt_lstm_bwd + qt_dense2.
 * gHn2 (optional, row stride ld_gh2): a second gradient of H', added to gHn on load; add0 (optional, (N, Cb) dense): added to
 * output plane 0 of part a -- the two sums autograd otherwise makes with elementwise launches when H' feeds the next layer AND
 * the next time step, and when the cell's input is also the head's residual operand (model/seq2seq.py:152-186). */
int qt_lstm_dgrad_blocks(int N);
int qt_lstm_bwd_dgrad(const float* gO, int ld_go, const float* gHn, int ld_gh, const float* gCn, int ld_gc,
                      const float* gates, const float* Cprev, int ld_c, const float* wc, const float* ln,
                      int N, const int32_t* n_dev, int h, float* gG, float* gCprev, float* part, int accumulate,
                      const float* Wrows, const void* Whi, const void* Wlo, int Kb, int Cb, int Cbb, float* out, float* outb,
                      int out_sm /* != 0: the data-gradient planes 1 .. Kb-1 leave slice-major (as qt_cheb_clip_bwd reads them) */,
                      const float* gHn2, int ld_gh2, const float* add0, void* stream);
/* x (n fp32) -> hi, lo (n bf16 each) with x ~ hi + lo (hi = round(x), lo = round(x - hi)) */
int qt_split_bf16(const float* x, int64_t n, void* hi, void* lo, void* stream);

/* The whole backward pass of one gate-cell use (hidden 8 / 16) in one persistent launch: qt_lstm_bwd's cell backward, the data
 * gradient of qt_lstm_bwd_dgrad (same planes) AND the weight gradient gW = [T_0 .. T_{K-1} | S]^T gG of the forward
 * operand (a0 .. S as in qt_dense_lstm), K (Ca + Cab) + Ks <= 128 rows.  gG itself is never written.  slab: (nslab >=
 * qt_lstm_fused_blocks(), Kt, 4h), zeroed by the caller before the first use of the weight in a pass: every launch ADDS the partial
 * weight gradient of workgroup b to slab[b]; qt_colsum over the slabs gives gW (fixed order: reproducible).  part: as
 * qt_lstm_bwd_dgrad, one row per workgroup.  Replaces (model/model.py:394-424 backward) k_lstm_bwd + two GEMMs. */
int qt_num_cus(void);
int qt_lstm_fused_blocks(void);
int qt_lstm_bwd_fused(const float* gO, int ld_go, const float* gHn, int ld_gh, const float* gCn, int ld_gc,
                      const float* gates, const float* Cprev, int ld_c, const float* wc, const float* ln,
                      int N, const int32_t* n_dev, int h, float* gCprev, float* part, int accumulate,
                      const float* Wrows, int Kb, int Cb, int Cbb, float* out, float* outb,
                      const float* a0, int lda0, const float* a_rest, const float* a0b, int lda0b, const float* a_restb,
                      int Ka, int Ca, int Cab, const float* S, int Ks, float* slab, int nslab, void* stream);

/* decoder head input, model/seq2seq.py:160-165: Z (N, hp) = [relu(LayerNorm_o(O)) | concat | 0...], hp >= h+1.
 * Zb != NULL: the same row as two matrices, Z (N, h) and Zb (N, hp - h).  O rows have stride ld_o floats (0 = h): the raw
 * output gate may be read in place from the saved gate activations, gates[:, 3h:4h]. */
int qt_head_fwd(const float* O, int ld_o, const float* ln_o /* (2,h) */, const float* concat /* (N) or NULL */,
                int N, const int32_t* n_dev, int h, int hp, float* Z, float* Zb, void* stream);
/* gZ / gZb laid out like Z / Zb; gO (N,h), gconcat (N) or NULL; part (nblk, 2h) partial sums of g_ln_o,
 * nblk = qt_lstm_bwd_blocks(N, h) */
int qt_head_bwd(const float* gZ, const float* gZb, const float* O, int ld_o, const float* ln_o, int N, const int32_t* n_dev,
                int h, int hp, float* gO, float* gconcat, float* part, int accumulate, void* stream);

/* out (N, sum widths) = [src_0 | src_1 | ...] for up to 8 row-strided fp32 sources (host arrays of nsrc device pointers,
 * widths and row strides, all multiples of 4): Z = [X | H] of GConvLSTM (model/model.py:394-424 feed X and H to separate
 * convolutions; here they share one Chebyshev pass) and the re-mesh state matrix. */
int qt_concat(const float* const* srcs, const int* widths, const int* lds, int nsrc, int N, const int32_t* n_dev,
              float* out, void* stream);

/* decoder input after a re-mesh, model/seq2seq.py:484-487: out (N, 4) = [val4[:, 0] | posfeat (N, 3)]; posfeat == NULL
 * writes [val4[:, 0], 0, 0, 0] (the gradient of the same op with respect to val4). */
int qt_decoder_input(const float* val4, int ld /* row stride of val4 in floats, 0 = 4 */, const float* posfeat, int N,
                     const int32_t* n_dev, float* out, void* stream);

/* One-column message aggregate with strided operands: out[i * ldo] = act(alpha (L^ x)_i + beta p_i + gamma q_i) on single
 * columns x[j * ldx], p[i * ldp], q[i * ldq] (p, q nullable); act = QT_ACT_NONE or QT_ACT_TANH_RES (tanh(drop_i v) + res[i * ldr],
 * drop nullable); pad4: the output row is written as the 16 bytes (v, 0, 0, 0).  ell: the first-four-edges array of
 * qt_edges_norm, or NULL.  Serves the Clenshaw recurrence of a ChebConv with ONE output channel (the decoder's fc_out2,
 * model/seq2seq.py:121) after its coefficient columns have been applied: u = z [w_0 w_1 w_2], y = u_0 + L^ (u_1 + 2 L^ u_2) - u_2. */
int qt_spmm1(const int32_t* rowptr, const int32_t* col, const float* nrm, const int32_t* ell, int N, const int32_t* n_dev,
             const float* x, int ldx, float alpha, const float* p, int ldp, float beta, const float* q, int ldq, float gamma,
             float* out, int ldo, int pad4, int act, const float* res, int ldr, const float* drop, void* stream);

/* backward of the qt_dense epilogue activations: G = gY * act'(Y) (QT_ACT_RELU, QT_ACT_TANH_RES with res / drop as in
 * qt_dense); gres (N, res_stride) or NULL receives the gradient of the residual operand (column 0 = gY[:, 0], rest 0).
 * gY2 (optional, laid out like gY): a second gradient of Y, added to gY on load (Y with two consumers: the decoder's output goes
 * to the loss and, re-meshed, into the next step's input, model/seq2seq.py:380-398). */
int qt_act_bwd(const float* gY, const float* Y, const float* res, int res_stride, const float* drop, int act, int N,
               const int32_t* n_dev, int Co, float* G, float* gres, const float* gY2, void* stream);

/* ---------------------------------------------------------------- edge-softmax attention (TransformerConv)
 * torch_geometric TransformerConv(heads=1, concat=False, beta=False, edge_dim=2, root_weight=True) as configured by
 * model/model.py:51 (third-party arithmetic, restated in oracle/): SURVEY.md 8(f) row 1.
 *   out_i = sum_{j->i} d_ij alpha_ij (v_j + e_ij) + skip_i,  alpha = softmax_j(q_i.(k_j + e_ij) / sqrt(c_real)),
 *   e_ij = We [angle(j->i), dist(j,i)]  (graph_functions.py:358-370), d_ij = inverted-dropout multiplier (keep, seed).
 * proj (N, ld): q | k | v | skip column blocks of C (multiple of 4, zero padded above c_real) columns each;
 * xy (N, 2) node centroids in edge-attribute units; selfloop (N) > 0 where get_adj emits the pair (i, i), or NULL;
 * incoming edges of i = CSR row i (symmetric adjacency).  stats (N, 2) = running max / sum of the softmax.
 * qt_attn_bwd: g (N, C) -> gproj (N, ld) [dq | dk | dv | g], part (qt_attn_blocks, 2*C) partials of dWe^T.
 */
int qt_attn_blocks(int N, int C);
/* eattr (E, 2): [angle, dist] of the message col[e] -> row(e) for every stored edge, computed once per mesh by
 * qt_attn_edge_attrs and passed to qt_attn_fwd / _bwd (required there; xy is kept in their signatures and unused). */
int qt_attn_edge_attrs(const int32_t* rowptr, const int32_t* col, const float* xy, int N, const int32_t* n_dev,
                       float* eattr, int32_t* rev /* optional (E): position of the transposed entry */, void* stream);
int qt_attn_fwd(const int32_t* rowptr, const int32_t* col, const float* xy, const float* eattr, const float* selfloop,
                const float* proj, int ld, const float* We, int C, int c_real, int N, const int32_t* n_dev,
                float keep, uint32_t seed, const uint32_t* seed_dev /* optional device-side step counter mixed into seed */,
                float* out, float* stats, int G /* heads, 0 = 1 */, int ld_o /* row stride of out, 0 = G C */,
                int64_t ps, int64_t hs, int64_t hs_o /* strides, see below; 0 = C, 4C, C: rows side by side */, void* stream);
int qt_attn_bwd(const int32_t* rowptr, const int32_t* col, const float* xy, const float* eattr, const float* selfloop,
                const float* proj, int ld, const float* We, int C, int c_real, int N, const int32_t* n_dev,
                float keep, uint32_t seed, const uint32_t* seed_dev, const float* g, int ld_g /* row stride of g, 0 = C */,
                const float* stats, const float* out /* the forward's output */, int ld_o, float* gproj, float* part,
                int accumulate /* bit 0: add into part; bit 1: g already IS the skip block of gproj (not stored again) */, const int32_t* rev, float* coef /* scratch (E + N, 2), see below */, int E,
                int G /* heads, 0 = 1 */, int gmod /* g holds gmod column blocks, head g reads block g % gmod; 0 = G */,
                int64_t ps, int64_t hs, int64_t hs_g, int64_t hs_o /* 0 = C, 4C, C, C */, void* stream);
/* Backward in two gather passes without atomics.  D_i = sum_e alpha_e t_e equals g_i . (out_i - skip_i) (the forward output is the
 * alpha-weighted sum of the messages), so the target pass knows it before its edge loop and leaves every message's final
 * coefficients (scale alpha (t - D), alpha d) at coef[rev[e]] -- the slot of the transposed entry, in the SOURCE's row -- and the
 * source pass reads its row's slots in order and gathers only the q_i / g_i rows. */
/* G > 1: G convolutions on the same mesh in one launch (the eight GraphConv stacks of a GConvLSTM, model/model.py:394-424,
 * run layer by layer).  Head g reads the column block [g 4C, (g+1) 4C) of the proj rows (ld >= G 4C) and We[g] ((G, C, 2)), and
 * writes the column block g C of the out rows; stats (G, N, 2), coef (G, E + N, 2), part (qt_attn_blocks, G, 2C), gproj
 * laid out like proj; every head draws its own dropout mask.  The results are those of G separate calls.
 * Strides in floats: ps between the q / k / v / skip blocks of a proj (gproj) row, hs between the heads of proj, hs_o / hs_g between
 * the heads of out / g.  One dense (N, C) plane per block and head is ld = C, ps = N C, hs = 4 N C (what qt_proj_group writes with
 * Kb = 4): a gathered k or v row is then one whole 128-byte line of a contiguous array. */

/* ---------------------------------------------------------------- gate-weight packing of stacked ChebConvs
 * A GraphConv stack applies its ChebConvs with no nonlinearity in between (model/model.py:59-97, :95-96), so the eight
 * stacks of a GConvLSTM (model/model.py:263-463) are Chebyshev series in weight space, composed layer by layer:
 *   M[k] = sum_{a,b} (delta(a+b, k) + delta(|a-b|, k)) / 2 * P[a] W[b],   orders Ka + K - 1.
 * qt_compose_step: one branch, natural layout -- series P0 (4, Ka, in, h) with bias series B0 (4, Kb0, h) through the layer
 *   P1 (4, K, h, h), B1 (4, h) -> P (4, Ka + K - 1, in, h), B (4, Kb0 + K - 1, h) (the layer's bias joins order 0).
 * qt_compose2: the LAST product of both branches (x: in = cin, h: in = h), written straight into the packed gate matrix
 *   qt_dense_lstm multiplies with: rows k*C + c for Z = [X (cin_pad) | H (h)], then pad4(Kb0 + K - 1) bias rows (x and h
 *   branches summed); columns gate-major i, f, c, o.  W1: variant with H, ((Ka+K-1)(cin_pad + h) + pad4(..), 4h); W0: without
 *   H; WT1 / WT0: their transposes; any of them may be NULL.  Two layers per stack: Ka = K, Kb0 = 1, no step.
 * The _bwd entries return the gradients of all inputs (compose2: from gW1 / gW0, either may be NULL). */
int qt_compose_step_fwd(const float* P0, const float* B0, const float* P1, const float* B1, int Ka, int Kb0, int K, int in,
                        int h, float* P, float* B, void* stream);
int qt_compose_step_bwd(const float* P0, const float* B0, const float* P1, const float* B1, int Ka, int Kb0, int K, int in,
                        int h, const float* gP, const float* gB, float* gP0, float* gB0, float* gP1, float* gB1,
                        void* stream);
int qt_compose2_fwd(const float* Px0, const float* Bx0, const float* Px1, const float* Bx1, const float* Ph0,
                    const float* Bh0, const float* Ph1, const float* Bh1, int Ka, int Kb0, int K, int cin, int cin_pad, int h,
                    float* W1, float* W0, float* WT1 /* optional transposes, (4h, rows) */, float* WT0, void* stream);
int qt_compose2_bwd(const float* Px0, const float* Bx0, const float* Px1, const float* Bx1, const float* Ph0,
                    const float* Bh0, const float* Ph1, const float* Bh1, int Ka, int Kb0, int K, int cin, int cin_pad, int h,
                    const float* gW1, const float* gW0, float* gPx0, float* gBx0, float* gPx1, float* gBx1,
                    float* gPh0, float* gBh0, float* gPh1, float* gBh1, void* stream);

/* clip_grad_norm_(max_norm) + one Adam step (model/mpnnlstm.py:251-257; torch.optim.Adam defaults: no weight decay, no amsgrad)
 * on ONE flat fp32 parameter vector p (n) with gradient g, moments m, v: g is scaled in place by min(1, max_norm / (|g| + 1e-6))
 * (max_norm <= 0: no clipping), *step is incremented on the device, stat[0] receives |g| before clipping.  lr_dev != NULL: the
 * learning rate is read on the device (a scheduler updates it in place; hipGraph replays follow), else lr_host. */
int qt_flat_adam(float* p, float* g, float* m, float* v, int n, int32_t* step, const float* lr_dev, float lr_host,
                 float beta1, float beta2, float eps, float max_norm, float* stat, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QTMPNN_H */
