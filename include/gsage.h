/*
 * gsage.h -- C ABI of libgsage.so: the MI355X (gfx950) GraphSAGE training hot path.
 *
 * The reference (celestice106/graphsage) has no FFI or operator registry: its hot
 * path is Python (`GraphSAGE.forward`, `SkipGramLoss.forward[_with_details]`) on
 * top of torch / torch_geometric ops.  This header is therefore the build's own
 * boundary; every entry point names the reference code whose
 * device work it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - sizes and counts are int64_t; CSR indices are int32_t (E, N < 2^31);
 *   - fp32 everywhere (the reference trains in fp32; parity bar 1e-5 relative);
 *   - calls are asynchronous launches on `stream` (a hipStream_t passed as void*);
 *   - the library never allocates or frees device memory (callers bring the
 *     workspaces; query their sizes with the *_workspace_bytes functions);
 *   - return value: 0 = ok, negative = gsage_status below.  Errors detected on the
 *     device (index out of range) are OR-ed into a caller-owned int32 status word.
 */
#ifndef GSAGE_H
#define GSAGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gsage_stream_t; /* hipStream_t */

enum gsage_status {
  GSAGE_OK = 0,
  GSAGE_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad enum        */
  GSAGE_ERR_UNSUPPORTED = -2,   /* shape outside what the kernel handles        */
  GSAGE_ERR_WORKSPACE = -3,     /* workspace too small                          */
  GSAGE_ERR_LAUNCH = -4,        /* hipGetLastError() != hipSuccess after launch */
  GSAGE_ERR_TOO_LARGE = -5      /* size beyond the fixed scratch of a *_noalloc entry point */
};

/* bits OR-ed into device status words */
enum gsage_device_flag {
  GSAGE_FLAG_INDEX_OUT_OF_RANGE = 1
};

enum gsage_epilogue {
  GSAGE_EPI_NONE = 0,          /* bare SAGEConv (graphsage.py:138)                */
  GSAGE_EPI_RELU = 1,          /* F.relu; dropout in eval mode (graphsage.py:134) */
  GSAGE_EPI_RELU_DROPOUT = 2,  /* F.relu + nn.Dropout(p), train (graphsage.py:134-135) */
  GSAGE_EPI_L2NORM = 3         /* F.normalize(p=2, dim=1, eps=1e-12) (graphsage.py:142) */
};

int gsage_version(void);
const char *gsage_error_string(int status);

/* ------------------------------------------------------------------ K1: CSR build
 * Replaces the per-call scatter bookkeeping inside PyG's `scatter(reduce='mean')`
 * (called from graphsage.py:133,138) and the degree loops at
 * src/walks/negative_sampler.py:76-79 / torch.bincount at src/utils/graph_utils.py:38-39.
 *
 *   edge_index : int64 [2,E] row-major; row 0 = source j, row 1 = target i
 *   rowptr[N+1], col[E]     : in-edges grouped by target, col = sources, edges of one
 *                             target in ascending edge order (stable) -- bit-exact vs
 *                             bincount + stable argsort
 *   rowptr_t[N+1], col_t[E] : the transpose (grouped by source, col_t = targets)
 *   inv_deg[N]              : 1.0f / max(in_degree, 1)
 *   long_rows / long_rows_t : ids of rows with more than gsage_long_row_threshold()
 *                             entries (capacity `long_cap` >= E/threshold + 1 each); n_long[0] and
 *                             n_long[1] receive their counts (for the hub-row pass of
 *                             the aggregation kernels).  May be NULL (then skipped).
 *   status                  : int32[1], OR-ed with GSAGE_FLAG_INDEX_OUT_OF_RANGE when an
 *                             index is outside [0,N); such edges are dropped.
 */
size_t gsage_csr_workspace_bytes(int64_t E, int64_t N);
int gsage_long_row_threshold(void);
int gsage_csr_build(const int64_t *edge_index, int64_t E, int64_t N,
                    int32_t *rowptr, int32_t *col, int32_t *rowptr_t, int32_t *col_t,
                    float *inv_deg, int32_t *long_rows, int32_t *long_rows_t,
                    int64_t long_cap, int32_t *n_long, int32_t *status, void *ws,
                    size_t ws_bytes, gsage_stream_t stream);

/* ------------------------------------------- K2 / K7: segment gather (mean and sum)
 * K2 replaces `x.index_select(0, ei[0])` + `scatter_add_` + divide inside PyG's
 * SAGEConv.propagate (graphsage.py:133,138; direction restated at layers.py:191-195).
 * K7 replaces the autograd backward of the same (index_add with atomics) by a gather
 * over the transposed CSR: deterministic, no atomics.
 *
 *   out[i,:] = (add_to ? add_to[i,:] : 0) + s_i * sum_{p in [rowptr[i],rowptr[i+1])} w_p * x[col[p],:]
 *     fwd (mean): s_i = inv_deg[i], w_p = 1                   (edge_weight = NULL)
 *     bwd       : s_i = 1, w_p = edge_weight[col[p]] or 1     (row_scale   = NULL)
 *   x, out, add_to: fp32 [N,F] row-major; add_to may alias out.
 *   long_rows/n_long: hub-row list from gsage_csr_build (NULL => every row by one wave).
 */
int gsage_segment_gather(const int32_t *rowptr, const int32_t *col,
                         const float *row_scale, const float *edge_weight,
                         const float *x, const float *add_to, int64_t N, int64_t F,
                         float *out, const int32_t *long_rows, const int32_t *n_long,
                         gsage_stream_t stream);

/* Same, with a hint: src_nz_mask holds one bit per row of x (bit r%32 of word r/32), and a
 * CLEAR bit promises that row r of x is all zero, so it is not fetched.  In the training step the
 * gradient that reaches the layer-2 backward is non-zero only on the rows the pair batch
 * touched (PyG's backward scatters those zero rows like any other: graphsage.py:138 under
 * autograd); gsage_project_bwd_ex produces the mask.  NULL => gsage_segment_gather.  The live
 * neighbours of a row are compacted before they are summed, so sums may differ from the
 * unmasked call in the last bit (the order is still fixed for a given mask). */
int gsage_segment_gather_skip(const int32_t *rowptr, const int32_t *col,
                              const float *row_scale, const float *edge_weight,
                              const float *x, const float *add_to, int64_t N, int64_t F,
                              float *out, const int32_t *long_rows, const int32_t *n_long,
                              const uint32_t *src_nz_mask, gsage_stream_t stream);

/* Same again, for the case where the rows of BOTH x and add_to outside a set are zero by promise
 * and may hold anything (gsage_project_bwd_rows does not write them): bit r of live_mask clear =>
 * row r of x and row r of add_to count as zero and are not fetched.  F must be 64 or 128, add_to
 * non-NULL (GSAGE_ERR_UNSUPPORTED otherwise: the mask is not a hint here). */
int gsage_segment_gather_live(const int32_t *rowptr, const int32_t *col,
                              const float *row_scale, const float *edge_weight,
                              const float *x, const float *add_to, int64_t N, int64_t F,
                              float *out, const int32_t *long_rows, const int32_t *n_long,
                              const uint32_t *live_mask, gsage_stream_t stream);

/* Weighted variant: out[i,:] = sum_{p in row i} edge_w[p] * x[col[p],:] with one weight per
 * CSR entry (same order as col).  Used by the deterministic loss backward; also the SpMM a
 * weighted graph would need. */
int gsage_segment_gather_ew(const int32_t *rowptr, const int32_t *col, const float *edge_w,
                            const float *x, int64_t N, int64_t F, float *out,
                            const int32_t *long_rows, const int32_t *n_long,
                            gsage_stream_t stream);

/* K2 for the layer BEHIND the input layer, recomputed from the input layer's operands.  The rows
 * PyG gathers there are h1[j] = dropout(relu(W_l1 . agg1[j] + b1 + W_r1 . x[j])) (graphsage.py:
 * 133-135 feeding the next SAGEConv's propagate): 64..256 floats each, yet functions of the 7
 * input features of j and their neighbour mean.  This entry point fetches the 64-byte row
 * [agg1[j] | x[j]] per neighbour instead of the 256 B..1 KB row of h1 (every fabric read is a
 * 128-byte line: half to an eighth of the line traffic, from a table of N x 64 bytes) and redoes
 * the 16 -> Fo projection on the matrix cores (fp16 MFMA on a two-way split of the fp32 operands,
 * both scaled into fp16's range by exact powers of two taken from max |u| and max |W|, fp32
 * accumulation: errors <= ~4e-7 of the largest output; an Inf or NaN feature makes NaN rows):
 *   out[i,:] = inv_deg[i] * sum_{p in row i} act(W_l1 . agg1[col[p]] + b1 + W_r1 . x8[col[p]])
 *   agg1, x8 : fp32 [N,8], column 7 of both ZERO (the 7 input features padded to 8; the bias
 *              rides in that column), w_l / w_r : fp32 [Fo,8] padded alike, b_l [Fo]
 *   epilogue : GSAGE_EPI_RELU or GSAGE_EPI_RELU_DROPOUT, Fo = 64, 128 or 256.  Training takes the
 *              zero pattern of the h1 that was STORED, so both agree on which elements exist: a
 *              row's keep-and-positive bits travel with its 64-byte row (64 columns: in its two
 *              padding columns; more: behind it, the row then takes 128 bytes of the same line).
 *              relu_sign = the bits gsage_project_fwd_ex left (used if its flag word says they
 *              were written), else h1 [N,Fo] is read for h1 > 0; one of the two must be given.
 *              Give BOTH unless the caller knows which forward kernel ran: with relu_sign alone
 *              and a flag word that says "not written" (unaligned operands or keep mask sent the
 *              forward to a kernel that leaves no bits) there is no zero pattern to take, every
 *              mask comes out empty and out is all zero (no fault; graphsage_amd always passes h1).
 *              Values are recomputed, i.e. equal to the stored ones up to fp32 rounding.
 *   out      : fp32 [N,Fo] row-major, or column-blocked (out_blocked, gsage_project_fwd_blocked)
 *   ws       : gsage_segment_gather_hidden1_workspace_bytes(N, E, Fo, epilogue) bytes (the rows, the
 *              weight planes, one partial row per 1 024 neighbours of a hub row).
 *   groups   : the graph's group index, built ONCE per CSR by gsage_hidden1_groups_build into
 *              gsage_hidden1_groups_bytes(N, E) bytes: every row's neighbours cut into groups of
 *              32 (the matrix instruction's row count) out of slots of 8, listed per chunk of 64 rows, so a wave
 *              walks its chunk without computing where the next group starts.  Rows with more than
 *              1 024 neighbours are listed apart, in pieces of 1 024 that other waves sum (added up
 *              in piece order: results do not depend on scheduling).  N < 2^26. */
size_t gsage_hidden1_groups_bytes(int64_t N, int64_t E);
int gsage_hidden1_groups_build(const int32_t *rowptr, int64_t N, int64_t E, void *groups,
                               size_t groups_bytes, gsage_stream_t stream);
int gsage_segment_gather_hidden1_supported(int64_t Fo, int epilogue);
size_t gsage_segment_gather_hidden1_workspace_bytes(int64_t N, int64_t E, int64_t Fo, int epilogue);
int gsage_segment_gather_hidden1(const void *groups, int64_t E, const int32_t *col,
                                 const float *inv_deg, const float *agg1, const float *x8, const float *w_l,
                                 const float *b_l, const float *w_r, int64_t N, int64_t Fo,
                                 int epilogue, float p, const float *h1, const uint64_t *relu_sign,
                                 float *out, int out_blocked, void *ws, size_t ws_bytes,
                                 gsage_stream_t stream);

/* The same with the row table handed in: u_rows [gsage_hidden1_u_rows(N), gsage_hidden1_u_row_bytes(Fo, epilogue) / 4]
 * as gsage_project_fwd_u left it while it computed the input layer (no pack pass over agg1 / x8 /
 * the sign bits; the workspace holds only the weight planes and the hub rows' partial sums). */
size_t gsage_hidden1_u_row_bytes(int64_t Fo, int epilogue);
int64_t gsage_hidden1_u_rows(int64_t N);   /* N + 1: row N's first word receives max |u| (float bits) */
size_t gsage_segment_gather_hidden1_u_workspace_bytes(int64_t E, int64_t Fo);
int gsage_segment_gather_hidden1_u(const void *groups, int64_t E, const int32_t *col,
                                   const float *inv_deg, const float *u_rows, const float *w_l,
                                   const float *b_l, const float *w_r, int64_t N, int64_t Fo,
                                   int epilogue, float p, float *out, int out_blocked, void *ws,
                                   size_t ws_bytes, gsage_stream_t stream);

/* ------------------------------------------------------------ K3: fused projection
 * Replaces `F.linear(agg, W_l, b_l) + F.linear(x, W_r)` (PyG SAGEConv.forward) fused
 * with F.relu / nn.Dropout (graphsage.py:134-135) or F.normalize (graphsage.py:141-142).
 *
 *   out[N,Fo] = epi( agg[N,Fi] . W_l[Fo,Fi]^T + b_l[Fo] + x[N,Fi] . W_r[Fo,Fi]^T )
 *   keep_mask : uint8 [N,Fo] (1 = keep) for RELU_DROPOUT, or NULL to draw the mask in
 *               the kernel from (seed, element index); kept values are scaled 1/(1-p).
 *   rnorm     : fp32 [N], written for L2NORM: 1 / max(||row||, 1e-12) (saved for bwd).
 *   wpack     : fp32 workspace, gsage_project_workspace_bytes(Fi,Fo) bytes.
 *   fp32-input MFMA (v_mfma_f32_32x32x2_f32) when Fi and Fo are in {64,128,256} (64 -> 64
 *   has a specialised persistent kernel with the weights in registers); a VALU kernel
 *   otherwise (e.g. the 8-wide first layer).  `force_generic`: 0 = choose by shape,
 *   1 = VALU kernel regardless, 2 = generic MFMA tiles even for 64 -> 64 (tests).
 */
size_t gsage_project_workspace_bytes(int64_t Fi, int64_t Fo);
int gsage_project_fwd(const float *agg, const float *x, const float *w_l,
                      const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                      int64_t Fo, int epilogue, const uint8_t *keep_mask,
                      uint64_t seed, float p, float *out, float *rnorm, void *wpack,
                      int force_generic, gsage_stream_t stream);

/* Backward of K3 (replaces autograd through linear / relu / dropout / normalize).
 *
 *   dpre[N,Fo]   = d(loss)/d(pre-epilogue), computed from grad_out, the saved `out`
 *                  and (L2NORM) `rnorm`; workspace owned by the caller.
 *   grad_agg[N,Fi] = (dpre . W_l) * (agg_scale ? agg_scale[row] : 1)   (NULL => skipped)
 *   grad_x[N,Fi]   =  dpre . W_r                                       (NULL => skipped)
 *   grad_w_l[Fo,Fi] = dpre^T . agg ; grad_w_r[Fo,Fi] = dpre^T . x ; grad_b[Fo] = sum_rows dpre
 *   ws: gsage_project_bwd_workspace_bytes(N,Fi,Fo) bytes (weight packs + partial slabs,
 *       reduced in a fixed order => bitwise reproducible).
 */
size_t gsage_project_bwd_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo);
int gsage_project_bwd(const float *grad_out, const float *out, const float *rnorm,
                      const float *agg, const float *x, const float *w_l,
                      const float *w_r, const float *agg_scale, int64_t N, int64_t Fi,
                      int64_t Fo, int epilogue, float p, float *dpre, float *grad_agg,
                      float *grad_x, float *grad_w_l, float *grad_b, float *grad_w_r,
                      void *ws, size_t ws_bytes, int force_generic,
                      gsage_stream_t stream);
/* The same pair with two optional side channels between a forward and its backward (both may be
 * NULL; neither changes a result):
 *   nz_mask   (bwd out): uint32 [ceil(N/32)], bit r%32 of word r/32 is CLEAR only if row r of dpre
 *             (hence of grad_agg and grad_x) is all zero.  The fused 64 -> 64 kernel tests the rows
 *             as it walks them; every other shape reports all rows as possibly non-zero.  Feeds
 *             gsage_segment_gather_skip.
 *   relu_sign (fwd out, bwd in): gsage_relu_sign_bytes(N, Fo) bytes.  For the RELU / RELU_DROPOUT
 *             epilogues the first-layer MFMA kernel (Fi = 8) leaves one bit per output element,
 *             set where out > 0, and the first-layer dW kernel reads those 8 bytes per row
 *             instead of the 256-byte row of `out` (autograd keeps and re-reads the whole
 *             activation for relu'/dropout': graphsage.py:134-135).  The last word says whether
 *             the bits were written; forward kernels that do not write them clear it, and the
 *             backward then reads `out` as usual. */
size_t gsage_relu_sign_bytes(int64_t N, int64_t Fo);
int gsage_project_fwd_ex(const float *agg, const float *x, const float *w_l,
                         const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                         int64_t Fo, int epilogue, const uint8_t *keep_mask,
                         uint64_t seed, float p, float *out, float *rnorm, void *wpack,
                         int force_generic, uint64_t *relu_sign, gsage_stream_t stream);
/* gsage_project_fwd_ex with a third side channel, forward to the NEXT layer's aggregation:
 *   u_rows (fwd out, nullable): fp32 [gsage_hidden1_u_rows(N), gsage_hidden1_u_row_bytes(Fo, epilogue) / 4].  The
 *             first-layer MFMA kernel (Fi = 8, RELU / RELU_DROPOUT, Fo = 64, 128 or 256) holds
 *             each row's [agg | x] and, in training, its keep-and-positive ballots: it writes the
 *             64-byte rows gsage_segment_gather_hidden1_u fetches (graphsage.py:133-135 feeding
 *             the next SAGEConv's propagate) while it is at it.  *u_written (HOST int, nullable)
 *             says whether the kernel that was dispatched wrote them (unaligned operands do not
 *             take that kernel); 0 = use gsage_segment_gather_hidden1, which packs the rows itself.
 *   amax_in / amax_out (nullable): as in gsage_project_fwd_range, here also for the training
 *             epilogues.  A 64 -> 64 layer that is given amax_in (max |x| as float bits; its agg is
 *             a mean of rows of x) runs on the fp16 matrix pipe with two-part operands scaled by
 *             that bound (three products per fp32 product, errors <= ~1e-6 of the largest
 *             pre-activation) instead of the fp32 MFMA kernel; without it the fp32 kernel runs. */
int gsage_project_fwd_u(const float *agg, const float *x, const float *w_l, const float *b_l,
                        const float *w_r, int64_t N, int64_t Fi, int64_t Fo, int epilogue,
                        const uint8_t *keep_mask, uint64_t seed, float p, float *out, float *rnorm,
                        void *wpack, int force_generic, uint64_t *relu_sign, float *u_rows,
                        int *u_written, const uint32_t *amax_in, uint32_t *amax_out,
                        gsage_stream_t stream);
int gsage_project_bwd_ex(const float *grad_out, const float *out, const float *rnorm,
                         const float *agg, const float *x, const float *w_l,
                         const float *w_r, const float *agg_scale, int64_t N, int64_t Fi,
                         int64_t Fo, int epilogue, float p, float *dpre, float *grad_agg,
                         float *grad_x, float *grad_w_l, float *grad_b, float *grad_w_r,
                         void *ws, size_t ws_bytes, int force_generic, uint32_t *nz_mask,
                         const uint64_t *relu_sign, gsage_stream_t stream);

/* The backward restricted to the rows a pair batch touched (gsage_touched_rows): autograd hands
 * the layer behind the loss a dense [N,Fo] gradient in which only those rows are non-zero
 * (loss.py:97-103 under autograd: three index_put(accumulate) into zeros), and PyG's backward
 * then pushes all N rows through the GEMMs.  Here the 32-row tiles are cut from the sorted list
 * `live_rows` (`*n_live` entries on the device, at most live_cap): rows outside it contribute
 * nothing to grad_w_l / grad_w_r / grad_b and are neither read nor written -- their rows of
 * grad_agg / grad_x are left as they are and must be masked by the consumer
 * (gsage_segment_gather_live).  Same results as gsage_project_bwd_ex on the listed rows and for
 * the weight gradients.  Only where gsage_project_bwd_rows_supported(Fi, Fo) (64 -> 64). */
int gsage_project_bwd_rows_supported(int64_t Fi, int64_t Fo);
int gsage_project_bwd_rows(const float *grad_out, const float *out, const float *rnorm,
                           const float *agg, const float *x, const float *w_l, const float *w_r,
                           const float *agg_scale, int64_t N, int64_t Fi, int64_t Fo, int epilogue,
                           float p, float *grad_agg, float *grad_x, float *grad_w_l, float *grad_b,
                           float *grad_w_r, void *ws, size_t ws_bytes, const int32_t *live_rows,
                           const int32_t *n_live, int64_t live_cap, gsage_stream_t stream);

/* gsage_project_bwd_rows that also publishes the range of what it wrote: *gmax_out (device word,
 * zeroed by the call) receives max |grad_agg|, |grad_x| over the listed rows as float bits, for the
 * fp16 scaling of gsage_hidden1_wgrad. */
int gsage_project_bwd_rows_range(const float *grad_out, const float *out, const float *rnorm,
                                 const float *agg, const float *x, const float *w_l, const float *w_r,
                                 const float *agg_scale, int64_t N, int64_t Fi, int64_t Fo,
                                 int epilogue, float p, float *grad_agg, float *grad_x,
                                 float *grad_w_l, float *grad_b, float *grad_w_r, void *ws,
                                 size_t ws_bytes, const int32_t *live_rows, const int32_t *n_live,
                                 int64_t live_cap, uint32_t *gmax_out, gsage_stream_t stream);

/* K7h: the INPUT layer's weight gradients straight from the backward of the layer behind it.
 * Replaces, under autograd, the backward of graphsage.py:133-138 between the second layer's
 * projection and the first layer's parameters: PyG scatters the gradient of layer 2's aggregation
 * to the neighbours (an [N,64] table, two fabric lines fetched per live edge), applies
 * ReLU'/dropout' of layer 1 and multiplies by layer 1's operands.  Only the weight gradients are
 * wanted there (the input is the feature matrix), so the sums are taken the other way round:
 *   dW1[c][k] = 1/(1-p) sum_{i listed} ( sum_{j in N(i)} m[j][c] grad_agg[i][c] uu[j][k]
 *                                        + m[i][c] grad_x[i][c] uu[i][k] ),   uu[j] = [agg1[j] | 1 | x[j] | 0]
 * -- per live edge ONE 64-byte row of the table the forward's recompute gather reads (u_rows: its
 * padding words hold the 64 keep-and-positive bits m[j] of the stored activation), nothing written.
 *   rowptr/col : the forward CSR (destination -> sources), N rows, E edges (N < 2^26)
 *   u_rows     : fp32 [N + 1, 16] as gsage_project_fwd_u left it (epilogue RELU_DROPOUT, Fo = 64)
 *   grad_agg   : fp32 [N,64], ALREADY scaled by inv_deg[row] (gsage_project_bwd_rows' agg_scale);
 *   grad_x     : fp32 [N,64]; of both only the listed rows are read
 *   g_amax     : max |grad_agg|, |grad_x| over the listed rows, float bits (gsage_project_bwd_rows_range)
 *   rows/n_rows/rows_cap : the ascending row list of gsage_touched_rows
 *   grad_w_l / grad_w_r : fp32 [64,8] (column 7, the padding, is set to 0), grad_b fp32 [64]
 *   ws         : gsage_hidden1_wgrad_workspace_bytes(rows_cap, E) bytes
 * Operands go through the fp16 matrix pipe as two-part values scaled by powers of two (as in
 * gsage_segment_gather_hidden1); sums are taken in an order fixed by the row list alone: bitwise
 * reproducible.  Only where gsage_hidden1_wgrad_supported(Fo, epilogue) (64, RELU_DROPOUT). */
int gsage_hidden1_wgrad_supported(int64_t Fo, int epilogue);
size_t gsage_hidden1_wgrad_workspace_bytes(int64_t rows_cap, int64_t E);
int gsage_hidden1_wgrad(const int32_t *rowptr, const int32_t *col, int64_t N, int64_t E,
                        const float *u_rows, const float *grad_agg, const float *grad_x,
                        const uint32_t *g_amax, const int32_t *rows, const int32_t *n_rows,
                        int64_t rows_cap, int64_t Fo, int epilogue, float p, float *grad_w_l,
                        float *grad_b, float *grad_w_r, void *ws, size_t ws_bytes,
                        gsage_stream_t stream);

/* K3 forward on COLUMN-BLOCKED activations (inference of the wide models, encoder.py:181-185 with
 * hidden_dim 128/256): a blocked [N,F] matrix is F/64 separate row-major [N,64] tables laid end
 * to end (element (r,c) at (c/64)*N*64 + r*64 + c%64).  The 256-wide gather over one 10 GB table
 * runs 15 % below the same bytes gathered from 2.56 GB tables, so a hidden layer
 * writes its output blocked, the next layer's aggregation is F/64 gathers of width 64
 * (gsage_segment_gather on each table), and its projection reads both operands blocked.
 * in_blocked / out_blocked say which side is blocked (agg and x share the input layout); same
 * arithmetic as gsage_project_fwd (the GEMM gives the same bits for the same inputs).  Eval epilogues only
 * (NONE, RELU, L2NORM); gsage_project_fwd_blocked_supported says which shapes have a kernel
 * (8 -> 64k row-major in / blocked out; 128|256 -> 128|256 on the split-operand GEMMs). */
int gsage_project_fwd_blocked_supported(int64_t Fi, int64_t Fo, int in_blocked, int out_blocked,
                                        int epilogue);
int gsage_project_fwd_blocked(const float *agg, const float *x, const float *w_l,
                              const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                              int64_t Fo, int epilogue, float *out, float *rnorm, void *wpack,
                              int in_blocked, int out_blocked, gsage_stream_t stream);

/* gsage_project_fwd_blocked with the RANGE of the activations handed from layer to layer (a chain
 * of wide layers in inference, encoder.py:181-185 with hidden_dim 256).  The 256-column forward GEMM
 * with K >= 256 runs on the fp16 matrix pipe: every fp32 operand as the sum of TWO fp16 numbers
 * (22 bits), three products per fp32 product, fp32 accumulation -- half the matrix work of the
 * bf16 scheme at the same accuracy (max error 5e-7 of the largest output; fp32 MFMA: 8e-7),
 * PROVIDED the operands are scaled into fp16's range first: weights per output column, the
 * activations by one power of two taken from max |agg|, |x|.  Tiles holding a row whose largest
 * element is more than 2^18 below that maximum (or above it: a bound that was too low) are
 * recomputed by the bf16 kernel; Inf/NaN or a maximum outside 2^+-40 sends every tile there.
 *   amax_in  : uint32 [1] or NULL: an upper bound of |agg| and |x| as IEEE-754 float bits (the
 *              mean aggregation of x is bounded by max |x|, so the producer's amax_out serves).
 *              NULL: the library finds the maximum itself (two streaming passes over agg and x,
 *              +30 % on the GEMM's time).  Ignored by kernels that do not scale.
 *   amax_out : uint32 [1] or NULL: receives max |out| as float bits (NaN patterns compare above
 *              Inf).  Must not be amax_in.  Only where gsage_project_fwd_range_supported(Fi, Fo,
 *              epilogue): the first-layer MFMA kernel (Fi = 8) and the 128/256-column tile kernels;
 *              GSAGE_ERR_UNSUPPORTED otherwise.
 * in_blocked = out_blocked = 0 is the plain row-major layout.  Eval epilogues only. */
int gsage_project_fwd_range_supported(int64_t Fi, int64_t Fo, int epilogue);
int gsage_project_fwd_range(const float *agg, const float *x, const float *w_l, const float *b_l,
                            const float *w_r, int64_t N, int64_t Fi, int64_t Fo, int epilogue,
                            float *out, float *rnorm, void *wpack, int in_blocked, int out_blocked,
                            const uint32_t *amax_in, uint32_t *amax_out, gsage_stream_t stream);

/* ---------------------------------------------------------- K4 / K5: skip-gram loss
 * K4 replaces SkipGramLoss.forward / forward_with_details (src/model/loss.py:97-136,
 * 161-193): three row gathers, B + B*K dots, x scale, logsigmoid means, and the
 * monitoring means, in one kernel with no [B,K,D] temporary.
 *
 *   emb [N,D] fp32; targets, contexts int64 [B]; negatives int64 [B,K]
 *   coef [B,1+K] fp32 out: d(loss)/d(raw score): coef[b,0] = -s*sigmoid(-s*pos_b)/B,
 *                          coef[b,1+k] = s*sigmoid(s*neg_bk)/(B*K)   (saved for K5)
 *   stats fp32[8] out: total_loss, pos_loss, neg_loss, pos_score_mean, neg_score_mean,
 *                      pos_prob, neg_prob (loss.py:185-193 order), and [7] = device flag
 *                      word (as float; non-zero => an index was out of range).
 *   loss_norm_B / loss_norm_BK: the mean denominators; pass B and B*K for the reference
 *   semantics, or the GLOBAL batch sizes when this call handles one shard of a batch
 *   (multi-GPU), so shard losses add up to the global mean.
 *   ws: gsage_skipgram_workspace_bytes(B) bytes of per-block partials.
 *
 * K5 replaces autograd's three index_put(accumulate=True) into a zero [N,D] tensor:
 *   grad_emb[t_b] += g*(coef_b0*e[c_b] + sum_k coef_bk*e[n_bk]); grad_emb[c_b] += g*coef_b0*e[t_b];
 *   grad_emb[n_bk] += g*coef_bk*e[t_b], g = *grad_loss (device scalar).
 *   grad_emb must be zero-filled by the caller (or hold a gradient to accumulate into).
 */
size_t gsage_skipgram_workspace_bytes(int64_t B);
int gsage_skipgram_fwd(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                       const int64_t *contexts, const int64_t *negatives, int64_t B,
                       int64_t K, float scale, double loss_norm_B, double loss_norm_BK,
                       float *coef, float *stats, void *ws, size_t ws_bytes,
                       gsage_stream_t stream);
int gsage_skipgram_bwd(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                       const int64_t *contexts, const int64_t *negatives, int64_t B,
                       int64_t K, const float *coef, const float *grad_loss,
                       float *grad_emb, gsage_stream_t stream);

/* The rows of the embedding table a pair batch names, i.e. the only rows where K5's gradient is
 * non-zero (loss.py:97-103).  mask: uint32 [ceil(N/32)], bit r%32 of word r/32 set iff row r is a
 * target, context or negative of the batch (out-of-range ids are ignored); rows: the set bits as
 * an ascending int32 list (first min(*n_rows, rows_cap) entries valid; rows_cap = min(N, (2+K)B)
 * always suffices); n_rows: int32 [1] on the device.  ws: gsage_touched_rows_workspace_bytes(N).
 * gsage_zero_rows clears exactly those rows of a [*, D] fp32 buffer (D % 4 == 0). */
size_t gsage_touched_rows_workspace_bytes(int64_t N);
int gsage_touched_rows(const int64_t *targets, const int64_t *contexts, const int64_t *negatives,
                       int64_t B, int64_t K, int64_t N, uint32_t *mask, int32_t *rows,
                       int64_t rows_cap, int32_t *n_rows, void *ws, size_t ws_bytes,
                       gsage_stream_t stream);
int gsage_zero_rows(const int32_t *rows, const int32_t *n_rows, int64_t rows_cap, int64_t D,
                    float *buf, gsage_stream_t stream);
/* The same, also leaving word_prefix[(N+31)/32] (nullable): the place in `rows` of each mask
 * word's first set bit, so that rank(r) = word_prefix[r >> 5] + popcount(mask[r >> 5] below bit
 * r & 31) for every listed row r (what gsage_skipgram_bwd_rows keys its terms by). */
int gsage_touched_rows_ex(const int64_t *targets, const int64_t *contexts,
                          const int64_t *negatives, int64_t B, int64_t K, int64_t N, uint32_t *mask,
                          int32_t *rows, int64_t rows_cap, int32_t *n_rows, int32_t *word_prefix,
                          void *ws, size_t ws_bytes, gsage_stream_t stream);

/* K4 for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers
 * that must not (or cannot) call hipMalloc or bring a workspace -- code under HIP graph capture,
 * tight serving loops that evaluate the loss, harnesses that forbid allocation.  Same arguments,
 * results and kernels as gsage_skipgram_fwd, minus `ws`: the per-workgroup partial sums live in a
 * 64 KiB file-scope device array of the library (one copy per GPU, held until the process ends),
 * whose address is looked up on the first call on each device (so make that first call outside a
 * graph capture).  The call makes no hipMalloc / hipFree and calls no allocating routine.
 *   limit : B <= gsage_skipgram_noalloc_max_pairs() (GSAGE_SKIPGRAM_NOALLOC_MAX_PAIRS = 2^24);
 *           larger B returns GSAGE_ERR_TOO_LARGE and launches nothing.
 *   All calls on one GPU share that GPU's array: calls that may OVERLAP on one GPU (different
 *   streams, different host threads) race on it -- serialise them or use gsage_skipgram_fwd with
 *   a workspace each.  Calls in stream order are safe: each launch rewrites every partial it
 *   later reads, so nothing has to be cleared between calls. */
#define GSAGE_SKIPGRAM_NOALLOC_MAX_PAIRS ((int64_t)1 << 24)
int64_t gsage_skipgram_noalloc_max_pairs(void);
int gsage_skipgram_fwd_noalloc(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                               const int64_t *contexts, const int64_t *negatives, int64_t B,
                               int64_t K, float scale, double loss_norm_B, double loss_norm_BK,
                               float *coef, float *stats, gsage_stream_t stream);

/* K5, deterministic form: the same gradient as gsage_skipgram_bwd without atomics.  The
 * 2(1+K) terms of every pair are stably sorted by destination row (K1's radix sort) into a
 * per-batch CSR and summed by the segment gather in a fixed order: bitwise reproducible.
 * grad_emb [N,D] is OVERWRITTEN (rows no pair touches become zero): no zero-fill needed.
 *   ws: gsage_skipgram_bwd_det_workspace_bytes(N, B, K) bytes.
 */
size_t gsage_skipgram_bwd_det_workspace_bytes(int64_t N, int64_t B, int64_t K);
int gsage_skipgram_bwd_det(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                           const int64_t *contexts, const int64_t *negatives, int64_t B,
                           int64_t K, const float *coef, const float *grad_loss,
                           float *grad_emb, void *ws, size_t ws_bytes, gsage_stream_t stream);

/* K5 without atomics over the TOUCHED ROWS ONLY (same autograd step of loss.py:97-103 as
 * gsage_skipgram_bwd_det).  The terms are keyed by the place of their destination row in the list
 * gsage_touched_rows_ex made of this batch (live_mask, word_prefix, rows, n_rows, rows_cap), so
 * the sort keys have log2(rows_cap) bits instead of log2 N, the per-batch CSR has one row per
 * listed row, and only the listed rows of grad_emb [N,D] are written -- each once, with its whole
 * sum, terms in pair order: bitwise reproducible.  Rows outside the list are NOT touched (they
 * are zero by definition; a consumer either masks them, gsage_segment_gather_live /
 * gsage_project_bwd_rows, or hands in a buffer that is zero there).  D = 64 or 128
 * (gsage_skipgram_bwd_rows_supported).  ws: gsage_skipgram_bwd_rows_workspace_bytes(rows_cap, B, K). */
int gsage_skipgram_bwd_rows_supported(int64_t D);
size_t gsage_skipgram_bwd_rows_workspace_bytes(int64_t rows_cap, int64_t B, int64_t K);
int gsage_skipgram_bwd_rows(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                            const int64_t *contexts, const int64_t *negatives, int64_t B,
                            int64_t K, const float *coef, const float *grad_loss, float *grad_emb,
                            const uint32_t *live_mask, const int32_t *word_prefix,
                            const int32_t *rows, const int32_t *n_rows, int64_t rows_cap, void *ws,
                            size_t ws_bytes, gsage_stream_t stream);

/* ----------------------------------------------------- feeder: random walks and pair windows
 * gsage_random_walks replaces RandomWalkGenerator.generate_all_walks (src/walks/generator.py:
 * 119-195): walk w starts at node w / walks_per_node, takes up to walk_length nodes, each step a
 * uniform choice among the out-neighbours (rowptr_t/col_t keep the reference's adjacency order),
 * stops at a dead end.  walks: int32 [N*walks_per_node, walk_length], padded with -1;
 * lengths: int32 [N*walks_per_node] (1 = isolated start; the reference drops those walks, they
 * yield no pairs).  The random stream is a counter hash of (seed, walk, step), not Python's.
 *
 * gsage_walk_pair_counts / gsage_walk_pairs_gather replace CooccurrencePairSampler.extract_pairs
 * (src/walks/pair_sampler.py:56-86) without materialising every pair: counts[w] = pairs of walk
 * w; with offsets = exclusive prefix sum of counts (W+1 entries, int64), pair number `id` in the
 * reference's enumeration order (walk, position i, window slot j) is decoded to
 * (walks[w][i], walks[w][j]).  pair_ids == NULL means ids first_id .. first_id+B-1 (bulk
 * extraction in reference order); otherwise any ids (uniform pair sampling for a batch).
 */
int gsage_random_walks(const int32_t *rowptr_t, const int32_t *col_t, int64_t N,
                       int64_t walk_length, int64_t walks_per_node, uint64_t seed,
                       int32_t *walks, int32_t *lengths, gsage_stream_t stream);
int gsage_walk_pair_counts(const int32_t *lengths, int64_t W, int64_t window, int64_t *counts,
                           gsage_stream_t stream);
int gsage_walk_pairs_gather(const int32_t *walks, const int32_t *lengths, int64_t walk_length,
                            const int64_t *offsets, int64_t W, const int64_t *pair_ids,
                            int64_t first_id, int64_t B, int64_t window, int64_t *targets,
                            int64_t *contexts, gsage_stream_t stream);

/* ------------------------------------------- feeder: alias table (HOST pointers, host code)
 * Replaces the interpreter loop of DegreeBiasedNegativeSampler._setup_alias_table
 * (src/walks/negative_sampler.py:95-145).  scaled_probs[i] = probs[i] * n (float32).
 * Sequential Vose construction in the reference's exact order (ascending fill, LIFO use,
 * float32 updates) => prob_table / alias are bit-identical to the reference's.
 * All three pointers are HOST memory; no GPU is touched.
 */
int gsage_alias_build_host(const float *scaled_probs, int64_t n, float *prob_table,
                           int64_t *alias);

/* ------------------------------------------------ feeder: negatives by the alias method
 * Replaces DegreeBiasedNegativeSampler.sample (src/walks/negative_sampler.py:147-178:
 * `idx = randint(N)`, `u = rand()`, `where(u >= prob_table[idx], alias[idx], idx)`) and the
 * per-batch call of it in BatchGenerator.__iter__ (src/training/batch_generator.py:181-187),
 * as ONE launch in front of the loss kernel (which reads `out` as its `negatives`).
 *   prob_table fp32 [N], alias int64 [N]: DEVICE copies of the table gsage_alias_build_host made;
 *   out int64 [count] (count = B*K, row-major [B,K]); draw q uses counter first_draw + q of the
 *   stream `seed` (a counter hash, not torch's Philox: draws agree in distribution, not value;
 *   the host mirror graphsage_amd/rng.py reproduces them bit for bit). */
int gsage_alias_sample(const float *prob_table, const int64_t *alias, int64_t N, int64_t count,
                       uint64_t seed, uint64_t first_draw, int64_t *out, gsage_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSAGE_H */
