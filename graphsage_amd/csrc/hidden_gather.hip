// K2h: mean aggregation of the FIRST hidden layer, recomputed from that layer's inputs.
//
// SAGEConv layer 2 aggregates h1[j] = act(W_l1 . agg1[j] + b1 + W_r1 . x[j]) over the neighbours j
// of every node (graphsage.py:133-135 feeding :133/:138 of the next layer).  A row of h1 is 64..256
// floats, but it is a function of 14 numbers: the 7 input features of j and their neighbour mean.
// The plain gather (aggregate.hip) fetches 256 B..1 KB per neighbour and runs at the fabric's
// 128-byte-line rate; this kernel fetches the 64-byte row u[j] = [agg1[j] | x[j]] instead (one
// line per neighbour, and a table a quarter to a sixteenth of the size) and redoes the 16 -> Fo
// projection on the matrix cores:
//
//   out[i,:] = inv_deg[i] * sum_{p in row i} act( [W_l1 | b1 | W_r1 | 0] . [agg1[j] | 1 | x[j] | 0] ),  j = col[p]
//
// A wave takes chunks of 64 destination rows (drawn from a work counter: the launch is as many waves
// as the chip holds); their neighbours are walked in groups of 32 (four slots of 8, see the group
// index below): the 32 rows u[j] are the A operand of
// v_mfma_f32_32x32x16_f16 (K = 16 is exactly the layer's input width, the bias rides in the
// padding column 7), every fp32 operand as the sum of TWO fp16 parts (22 bits of significand) and
// three of the four products issued, smallest first (project.hip, gemm_half_k: max error 4e-7 of
// the largest output, against 3e-7 for the three-way bf16 split used until round 3, at half the
// matrix instructions).  fp16 has the significand but not the range, so both operands are scaled
// by exact powers of two: the rows by 2^13 / A', A' = the power of two at or below max |u| over
// the whole table (which the kernel that writes the table publishes behind it), the weights by
// 2^13 over their own largest magnitude (h1_pack_w_k), undone by one factor at the row store.
// Absolute error of an operand <= max(2^-21 |v|, 2^-38 max|v|): results are accurate relative to
// the largest terms of a pre-activation, whatever the scale of the features.
// The accumulator comes out as D[neighbour][column] with the neighbours in the
// registers of a lane, so the activation and the sum over the neighbours are in-lane work
// and only the two lane halves are combined once per destination row.
//   eval  (RELU):         act = max(., 0)
//   train (RELU_DROPOUT): act = keep-and-positive bit of (j, column) ? . / (1 - p) : 0, the bits
//                         being those of the h1 that was stored (its ReLU'/dropout' bits, or
//                         h1 > 0), so the recomputed rows have the stored rows' zero pattern.
// The recomputed pre-activation differs from the stored one by fp32 rounding (1e-7 relative).
#include "gsage_common.h"
#include "gsage_sort.h"

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));      // (a B operand as stored: 8 halves)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

#ifndef H1_ROWS_PER_WAVE
#define H1_ROWS_PER_WAVE 64
#endif
#ifndef H1_WG_WAVES_
#define H1_WG_WAVES_ 1
#endif
constexpr int H1_WG_WAVES = H1_WG_WAVES_;   // waves per workgroup of the gather kernel (they share nothing)
constexpr int H1_ROWS = H1_ROWS_PER_WAVE;   // destination rows per wave, at most 64 (one trip per wave, in-order dispatch)

// ---- weight planes.  With A' = the power of two at or below max |u| (1 if that is 0, Inf or NaN):
//   W'[n][k] = k < 7 ? W_l[n][k] A' : k == 7 ? b[n] : k < 15 ? W_r[n][k-8] A' : 0,
// so that (u / A') . W' with a 1 in slot 7 is the pre-activation; then W' s_w, s_w = the power of
// two that takes max |W'| into [2^13, 2^14), split into two fp16 planes [2][Fo][16].
// params[0] = 2^13 / A' (the rows' scale), params[1] = 2^-13 / s_w (undoes both).  One workgroup.
constexpr float H1_TOP = 8192.f;      // 2^13: where the largest operand magnitudes are put
__device__ __forceinline__ float h1_pow2_floor(float v, float fallback) {
  // 2^floor(log2 v) for a normal positive v within 2^+-100, else `fallback`
  const uint32_t e = (__float_as_uint(v) >> 23) & 255u;
  if (!(v > 0.f) || e < 27u || e > 227u) return fallback;
  return __uint_as_float(e << 23);
}
__global__ void __launch_bounds__(256) h1_pack_w_k(const float *__restrict__ wl, const float *__restrict__ bl,
                                                   const float *__restrict__ wr, int Fo,
                                                   const uint32_t *__restrict__ u_amax,
                                                   uint16_t *__restrict__ planes, float *__restrict__ params) {
  __shared__ float red[4];
  const int total = Fo * 16;
  const float Ap = h1_pow2_floor(__uint_as_float(*u_amax), 1.f);
  auto value = [&](int e) {
    const int n = e >> 4, k = e & 15;
    return k < 7 ? wl[n * 8 + k] * Ap : k == 7 ? bl[n] : k < 15 ? wr[n * 8 + (k - 8)] * Ap : 0.f;
  };
  float m = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) m = fmaxf(m, fabsf(value(e)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sw = H1_TOP / h1_pow2_floor(m, H1_TOP);          // (no usable maximum: scale 1)
  for (int e = threadIdx.x; e < total; e += 256) {
    const float v = value(e) * sw;
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    planes[e] = __builtin_bit_cast(uint16_t, hi);
    planes[total + e] = __builtin_bit_cast(uint16_t, lo);
  }
  if (threadIdx.x == 0) {
    params[0] = H1_TOP / Ap;
    params[1] = 1.f / (H1_TOP * sw);
  }
}

// ---- the 64-byte rows: u[j] = [agg1[j][0..6], m_lo, x[j][0..6], m_hi].  m = the 64 keep-and-positive
// bits of columns 0..63 of row j (training; the masks of columns 64.. follow the row, which then
// takes 128 bytes), bit 16 cc + q of a 64-bit mask = column 4 q + cc of its group: the order
// the first-layer kernel's ballots come in (gsage_relu_sign_bytes: per 4 rows and column block four
// 64-bit words, word cc bit l = column 4 (l & 15) + cc of row l >> 4), so a row's mask is the four
// 16-bit fields of its group's words.  No sign bits (other forward kernels): taken from h1 > 0.
__global__ void __launch_bounds__(256) h1_pack_u_k(const float *__restrict__ agg1, const float *__restrict__ x8,
                                                   int64_t N, int train, int ngroups, const float *__restrict__ h1,
                                                   const unsigned long long *__restrict__ sign,
                                                   float *__restrict__ U, uint32_t *__restrict__ u_amax) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // the flag word behind the bits says whether the forward kernel that ran wrote them
  const int64_t ngrp4 = (N + 3) >> 2;
  const bool have_bits = train && sign && sign[ngrp4 * 4 * ngroups] != 0ull;
  f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
  if (j < N) {
    a0 = *(const f32x4v *)(agg1 + j * 8); a1 = *(const f32x4v *)(agg1 + j * 8 + 4);
    b0 = *(const f32x4v *)(x8 + j * 8); b1 = *(const f32x4v *)(x8 + j * 8 + 4);
  }
  {
    // max |u| over the table, for the gather's fp16 scaling (float bits of |v| compare as integers)
    uint32_t mx = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mx = max(mx, __float_as_uint(a0[e]) & 0x7FFFFFFFu); mx = max(mx, __float_as_uint(b0[e]) & 0x7FFFFFFFu);
      if (e < 3) { mx = max(mx, __float_as_uint(a1[e]) & 0x7FFFFFFFu); mx = max(mx, __float_as_uint(b1[e]) & 0x7FFFFFFFu); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > __atomic_load_n(u_amax, __ATOMIC_RELAXED)) atomicMax(u_amax, mx);
  }
  if (j >= N) return;
  // the mask of columns 64 g .. 64 g + 63 (group g), as two words
  auto mask_of = [&](int g, uint32_t &lo, uint32_t &hi) {
    lo = 0u; hi = 0u;
    if (have_bits) {
      const unsigned long long *w = sign + ((int64_t)g * ngrp4 + (j >> 2)) * 4;
      const int sh = 16 * (int)(j & 3);
      lo = (uint32_t)((w[0] >> sh) & 0xFFFFull) | ((uint32_t)((w[1] >> sh) & 0xFFFFull) << 16);
      hi = (uint32_t)((w[2] >> sh) & 0xFFFFull) | ((uint32_t)((w[3] >> sh) & 0xFFFFull) << 16);
    } else if (h1) {      // (neither source: the mask stays empty and the row contributes nothing)
      const float *r = h1 + j * (64 * ngroups) + 64 * g;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4v v = *(const f32x4v *)(r + 4 * q);
        lo |= (v[0] > 0.f ? 1u : 0u) << q | (v[1] > 0.f ? 1u : 0u) << (16 + q);
        hi |= (v[2] > 0.f ? 1u : 0u) << q | (v[3] > 0.f ? 1u : 0u) << (16 + q);
      }
    }
  };
  uint32_t lo = 0u, hi = 0u;
  if (train) mask_of(0, lo, hi);
  a1[3] = __uint_as_float(lo);
  b1[3] = __uint_as_float(hi);
  // rows of 64 bytes; training on more than 64 columns: 128 bytes (still one line per neighbour),
  // the masks of the further column groups behind the row, low words at byte 64, high words at 80
  float *u = U + j * (train && ngroups > 1 ? 32 : 16);
  *(f32x4v *)(u) = a0;
  *(f32x4v *)(u + 4) = a1;
  *(f32x4v *)(u + 8) = b0;
  *(f32x4v *)(u + 12) = b1;
  if (train && ngroups > 1) {
    u32x4v xl = {0u, 0u, 0u, 0u}, xh = {0u, 0u, 0u, 0u};
    for (int g = 1; g < ngroups; ++g) {
      uint32_t l, h_;
      mask_of(g, l, h_);
      xl[g - 1] = l;
      xh[g - 1] = h_;
    }
    *(u32x4v *)(u + 16) = xl;
    *(u32x4v *)(u + 20) = xh;
  }
}

// ---- group index, cut once per graph.  The matrix instruction takes 32 neighbours at a time, a
// row of the benchmark graphs has 11 at the median: one group per row would leave half the
// instruction's rows empty.  So a row's neighbours are cut into SLOTS of 8, and a group is any 4
// consecutive slots of a chunk of 64 destination rows (one wave), whatever rows they belong to: the
// accumulator registers 4 s .. 4 s + 3 of a lane (both lane halves) are exactly the 8 neighbours of
// slot s, so a row's sum is taken over register quads.  A slot descriptor is (p = first position in
// col, meta = count 0..8 | row in chunk << 4 | last slot of its row << 10 | valid << 11); a row
// without neighbours has one slot with count 0 (it stores its zero row).  Chunk k owns the groups
// goff[k] .. goff[k+1], its last group padded with invalid slots.
//
// HUB rows (more than H1_HUB_T neighbours) stay out of their chunk's list: a chunk is walked by one
// wave, and a 50 000-neighbour row in it keeps that wave busy for milliseconds while the rest of
// the launch has long finished (measured at C4: 5.4 ms, against 4.5 ms on the same graph with the
// degrees capped at 512).  A hub row is cut into PIECES of H1_PIECE neighbours (32 groups, about
// the length of an ordinary chunk); the pieces are the first items the launch's waves draw from
// their work counter -- so the long rows overlap everything else -- and each leaves one unscaled
// partial row, which h1_hub_combine_k adds up in piece order (fixed => reproducible) and scales.
// A piece's slot descriptors are not stored: they follow from (first position, length).
constexpr int H1_SLOT = 8;
constexpr int H1_META_LAST = 1 << 10, H1_META_VALID = 1 << 11;
#ifndef H1_HUB_T_
#define H1_HUB_T_ 1024
#endif
#ifndef H1_PIECE_
#define H1_PIECE_ 1024
#endif
#ifndef H1_GRID_WAVES_
#define H1_GRID_WAVES_ 4096
#endif
constexpr int H1_HUB_T = H1_HUB_T_;
constexpr int H1_PIECE = H1_PIECE_;              // a multiple of 32 (whole groups)
constexpr int H1_GRID_WAVES = H1_GRID_WAVES_;    // waves of the gather launch: 256 CUs x 4 SIMDs x 4 (what 128 registers allow)

__device__ __forceinline__ int h1_row_slots(int deg) {
  return deg > H1_HUB_T ? 0 : deg > 0 ? (deg + H1_SLOT - 1) / H1_SLOT : 1;
}

// hub rows -> hubs[h] = (row, first piece, pieces, 0), pieces[i] = (first position in col, length);
// cnt[0] = pieces, cnt[1] = hub rows (zeroed before the launch).  The order the rows land in is
// not fixed (atomics), the results do not depend on it: a row's pieces are consecutive and in order.
__global__ void __launch_bounds__(256) h1_hub_list_k(const int *__restrict__ rowptr, int64_t N,
                                                     int *__restrict__ cnt, int4 *__restrict__ hubs,
                                                     int64_t hub_cap, int2 *__restrict__ pieces,
                                                     int64_t piece_cap) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= N) return;
  const int p0 = rowptr[r], deg = rowptr[r + 1] - p0;
  if (deg <= H1_HUB_T) return;
  const int np = (deg + H1_PIECE - 1) / H1_PIECE;
  const int base = atomicAdd(cnt, np), h = atomicAdd(cnt + 1, 1);
  if (h >= hub_cap || (int64_t)base + np > piece_cap) return;     // cannot happen (caps are bounds)
  hubs[h] = make_int4((int)r, base, np, 0);
  for (int k = 0; k < np; ++k) {
    const int left = deg - k * H1_PIECE;
    pieces[base + k] = make_int2(p0 + k * H1_PIECE, left < H1_PIECE ? left : H1_PIECE);
  }
}

__global__ void __launch_bounds__(256) h1_groups_count_k(const int *__restrict__ rowptr, int64_t N,
                                                         int64_t nchunks, int *__restrict__ counts) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k > nchunks) return;
  int slots = 0;
  if (k < nchunks) {
    const int64_t r0 = k * H1_ROWS, r1 = r0 + H1_ROWS < N ? r0 + H1_ROWS : N;
    int prev = rowptr[r0];
    for (int64_t r = r0; r < r1; ++r) {
      const int nxt = rowptr[r + 1];
      slots += h1_row_slots(nxt - prev);
      prev = nxt;
    }
  }
  counts[k] = (slots + 3) >> 2;           // counts[nchunks] = 0: the scan leaves the total there
}

__global__ void __launch_bounds__(256) h1_groups_fill_k(const int *__restrict__ rowptr, int64_t N,
                                                        int64_t nchunks, const int *__restrict__ goff,
                                                        int2 *__restrict__ gdesc) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= nchunks) return;
  const int64_t r0 = k * H1_ROWS, r1 = r0 + H1_ROWS < N ? r0 + H1_ROWS : N;
  int2 *d = gdesc + (int64_t)goff[k] * 4;
  int2 *const end = gdesc + (int64_t)goff[k + 1] * 4;
  int prev = rowptr[r0];
  for (int64_t r = r0; r < r1; ++r) {
    const int nxt = rowptr[r + 1], deg = nxt - prev, rr = (int)(r - r0);
    const int ns = h1_row_slots(deg);
    for (int q = 0; q < ns; ++q) {
      const int left = deg - q * H1_SLOT;
      *d++ = make_int2(prev + q * H1_SLOT, (left < H1_SLOT ? (left > 0 ? left : 0) : H1_SLOT) | (rr << 4) |
                                           (q == ns - 1 ? H1_META_LAST : 0) | H1_META_VALID);
    }
    prev = nxt;
  }
  while (d < end) *d++ = make_int2(0, 0);
}

struct H1Groups {        // carved from the caller's buffer (gsage_hidden1_groups_bytes)
  int *goff;             // nchunks + 1 (+ scan block sums behind it)
  int *bsum;
  int2 *gdesc;           // 4 slots per group
  int *hcnt;             // [0] pieces, [1] hub rows
  int4 *hubs;
  int2 *pieces;
  int64_t hub_cap, piece_cap;
  size_t bytes;
};
static inline int64_t h1_chunks(int64_t N) { return (N + H1_ROWS - 1) / H1_ROWS; }
// slots <= N + E / 8 (a row of degree d > 0 has ceil(d / 8) <= 1 + d / 8 of them... minus: d / 8 + 1),
// groups <= slots / 4 + one partly filled group per chunk
static inline int64_t h1_max_groups(int64_t N, int64_t E) { return (N + E / H1_SLOT) / 4 + h1_chunks(N) + 1; }
static inline H1Groups h1_carve(void *buf, int64_t N, int64_t E) {
  GsCarver c(buf);
  H1Groups g;
  const int64_t nch = h1_chunks(N);
  g.goff = c.take<int>((size_t)nch + 1);
  g.bsum = c.take<int>((size_t)((nch + 1 + GS_SCAN_CH - 1) / GS_SCAN_CH + 1));
  g.gdesc = c.take<int2>((size_t)(4 * h1_max_groups(N, E)));
  // a hub row has more than H1_HUB_T neighbours and ceil(deg / H1_PIECE) <= deg / H1_PIECE + 1 pieces
  g.hub_cap = E / H1_HUB_T + 1;
  g.piece_cap = E / H1_PIECE + g.hub_cap;
  g.hcnt = c.take<int>(4);
  g.hubs = c.take<int4>((size_t)g.hub_cap);
  g.pieces = c.take<int2>((size_t)g.piece_cap);
  g.bytes = c.off;
  return g;
}
static inline int64_t h1_piece_cap(int64_t E) { return E / H1_PIECE + E / H1_HUB_T + 1; }

#ifndef H1_OCC_
#define H1_OCC_ 4
#endif
template <int NB, bool TRAIN>
__global__ void __launch_bounds__(64 * H1_WG_WAVES, H1_OCC_)
hidden1_gather_k(const int *__restrict__ goff, const int2 *__restrict__ gdesc,
                 const int *__restrict__ col, const float *__restrict__ inv_deg,
                 const float *__restrict__ U, const uint16_t *__restrict__ Wp,
                 const float *__restrict__ params, int64_t N, float scale,
                 float *__restrict__ out, int64_t out_blk, const int2 *__restrict__ pieces,
                 const int *__restrict__ hcnt, float *__restrict__ part, int *__restrict__ counter) {
  // bytes per row of the table: training on more than 64 columns carries the further column
  // groups' masks behind the 64-byte row (h1_pack_u_k)
  constexpr uint32_t UROW = (TRAIN && NB > 2) ? 128u : 64u;
  const int lane = gs_lane(), c = lane & 31, h = lane >> 5;
  // The launch is as many waves as the chip holds; they draw their work from one counter: first
  // the pieces of the hub rows (so the long rows overlap everything else), then the chunks of
  // H1_ROWS rows in ascending order (the chip works on a compact moving window of col / out, as
  // with in-order dispatch of one wave per chunk, but no wave slot waits for the dispatcher).
  // Which wave computes an item has no influence on the item's result.
  const int n_pieces = hcnt[0];
  const float su = params[0];            // rows -> fp16 range (2^13 / A')
  const float unscale = params[1];       // undoes the rows' and the weights' scales
  // B operand: column 32 b + c, k = 8 h .. 8 h + 7 of the two weight planes, for the whole launch
  f16x8 wf[NB][2];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      wf[b][pl] = __builtin_bit_cast(f16x8, *(const bf16x8 *)(Wp + ((int64_t)pl * 32 * NB + 32 * b + c) * 16 + 8 * h));

  // training: which mask word a lane reads (columns with c & 2: the high word, held by the lanes
  // of the upper half) and which bit of it per column block
  const int bp_base = 4 * (((c >> 1) & 1) * 32 + 4 * h);
  const int bit0 = (c & 1) * 16 + (c >> 2);
  const uint32_t u_lane = 32u * (uint32_t)h;       // byte offset of this lane's half of a 64-byte row
  const int sl = c >> 3, ln = c & 7;               // this lane's slot of a group, its place in the slot

  for (;;) {
    // ---- this wave's item: a chunk (rows row0 .. row0 + H1_ROWS, groups g0 .. g1 of the index) or a
    // piece (one run of `plen` neighbours from position pfirst of col, groups 0 .. g1)
    int drawn = 0;
    if (lane == 0) drawn = atomicAdd(counter, 1);
    drawn = __builtin_amdgcn_readfirstlane(drawn);
    const bool hub = drawn < n_pieces;
    const int64_t item = hub ? drawn : drawn - n_pieces;
    if (!hub && item * H1_ROWS >= N) return;
    // a piece's partial row is stored like row `piece` of an unblocked [pieces, 32 NB] table
    float *const o_base = hub ? part : out;
    const int64_t o_blk = hub ? 0 : out_blk;
    const int64_t row0 = hub ? item : item * H1_ROWS;
    int g0 = 0, g1, pfirst = 0, plen = 0;
    float idg = 0.f;                                 // lane k: the scale of row k of the chunk
    if (hub) {
      const int2 pc = pieces[item];
      pfirst = __builtin_amdgcn_readfirstlane(pc.x);
      plen = __builtin_amdgcn_readfirstlane(pc.y);
      g1 = (plen + 31) >> 5;
    } else {
      g0 = goff[item];
      g1 = goff[item + 1];
      idg = row0 + lane < N && lane < H1_ROWS ? inv_deg[row0 + lane] : 0.f;
    }
    float s[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = 0.f;

    // Three loads hang on one another -- slot descriptor, neighbour id, the row u[j] -- so the loop
    // runs them as a pipeline: while group g is computed, the rows of g + 1, the ids of g + 2 and
    // the descriptors of g + 3 are in flight.  (A piece's descriptors are arithmetic.)
#define H1_LOAD_DESC(DST, G)                                                                  \
  {                                                                                           \
    DST = make_int2(0, 0);                                                                    \
    if (hub) {                                                                                \
      const int left_ = plen - 8 * (4 * (G) + sl);                                            \
      if (left_ > 0)                                                                          \
        DST = make_int2(pfirst + 8 * (4 * (G) + sl),                                          \
                        (left_ < 8 ? left_ : 8) | (left_ <= 8 ? H1_META_LAST : 0) | H1_META_VALID); \
    } else if ((G) < g1) {                                                                    \
      DST = gdesc[(int64_t)(G) * 4 + sl];                                                     \
    }                                                                                         \
  }
#define H1_LOAD_COL(DST, DESC)                                                                \
  {                                                                                           \
    DST = -1;                                                                                 \
    if (ln < ((DESC).y & 15)) DST = col[(DESC).x + ln];                                       \
  }
#define H1_LOAD_U(A, B, X, ID)                                                                \
  {                                                                                           \
    A = (f32x4v){0.f, 0.f, 0.f, 0.f};                                                         \
    B = (f32x4v){0.f, 0.f, 0.f, 0.f};                                                         \
    X = (u32x4v){0u, 0u, 0u, 0u};                                                             \
    if ((ID) >= 0) {                                                                          \
      const char *p_ = (const char *)U + ((uint32_t)(ID) * UROW + u_lane);                    \
      A = *(const f32x4v *)p_;                                                                \
      B = *(const f32x4v *)(p_ + 16);                                                         \
      if (UROW > 64u) X = *(const u32x4v *)(p_ + (64 - 16 * h));   /* byte 64 + 16 h of the row */ \
    }                                                                                         \
  }
    int2 d0, d1, d2, d3;          // descriptors of group g, g + 1, g + 2, g + 3 (this lane's slot)
    int id1, id2;                 // neighbour ids of g + 1, g + 2
    f32x4v ua, ub;                // the row of g
    u32x4v ux;                    // (wide training) the masks of its column groups 1, 2, 3
    bool live;
    {
      int id0;
      H1_LOAD_DESC(d0, g0)
      H1_LOAD_DESC(d1, g0 + 1)
      H1_LOAD_DESC(d2, g0 + 2)
      H1_LOAD_COL(id0, d0)
      H1_LOAD_COL(id1, d1)
      H1_LOAD_U(ua, ub, ux, id0)
      live = id0 >= 0;
    }
    for (int g = g0; g < g1; ++g) {
      f32x4v na, nb;
      u32x4v nx;
      H1_LOAD_U(na, nb, nx, id1)
      H1_LOAD_COL(id2, d2)
      H1_LOAD_DESC(d3, g + 3)
      // ---- this group: A operand = row of neighbour c, k = 8 h .. 8 h + 7
      uint32_t mword[NB / 2];
#pragma unroll
      for (int g_ = 0; g_ < NB / 2; ++g_) mword[g_] = 0u;
      if (TRAIN) {
        mword[0] = live ? __float_as_uint(ub[3]) : 0u;
#pragma unroll
        for (int g_ = 1; g_ < NB / 2; ++g_) mword[g_] = ux[g_ - 1];     // zero without a neighbour
      }
      // scaled into fp16 range and cut in two: hi = rtz16(v), lo = rtz16(v - hi) (the difference is
      // exact).  Slot k = 7 is the bias column (the value 1, i.e. 2^13 after scaling; 0 for a lane
      // without a neighbour), k = 15 carries nothing.
      f16x8 a1, a2;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        float v0 = (e < 4 ? ua[e] : ub[e - 4]) * su, v1 = (e + 1 < 4 ? ua[e + 1] : ub[e - 3]) * su;
        if (e == 6) v1 = (live && h == 0) ? H1_TOP : 0.f;
        const f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0, v1));
        const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0 - (float)hi[0], v1 - (float)hi[1]));
        a1[e] = hi[0]; a1[e + 1] = hi[1];
        a2[e] = lo[0]; a2[e + 1] = lo[1];
      }
      // 64 columns (two blocks) at a time, so that the accumulators and slot sums of a wide layer
      // never live together
#pragma unroll
      for (int bp = 0; bp < NB / 2; ++bp) {
        float t[2][4];              // per column block: the sums over the 8 neighbours of slot 0..3
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int b = 2 * bp + bb;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          // smallest terms first, so that they are not rounded away against the big partial sums
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, wf[b][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[b][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, wf[b][0], acc, 0, 0, 0);
          // acc[r] = pre-activation of neighbour (r & 3) + 8 (r >> 2) + 4 h, column 32 b + c
          float v[16];
          if (TRAIN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int nb_ = (r & 3) + 8 * (r >> 2);
              const uint32_t m = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + 4 * nb_, (int)mword[bp]);
              const int keep = __builtin_amdgcn_sbfe((int)m, bit0 + 8 * bb, 1);    // 0 or -1
              v[r] = __uint_as_float((uint32_t)keep & __float_as_uint(acc[r]));
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r], 0.f);      // one v_max (-fno-honor-nans)
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) t[bb][q] = (v[4 * q] + v[4 * q + 1]) + (v[4 * q + 2] + v[4 * q + 3]);
        }
        // ---- hand the slot sums to their rows; a row's last slot combines the lane halves and stores
        // (a piece: its one "row" is the partial of the piece, left unscaled)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int meta = __builtin_amdgcn_readlane(d0.y, 8 * q);
          if (meta & H1_META_VALID) {
            s[2 * bp] += t[0][q];
            s[2 * bp + 1] += t[1][q];
            if (meta & H1_META_LAST) {
              const int r = (meta >> 4) & 63;
              const float f = hub ? unscale
                                  : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, idg), r)) * (scale * unscale);
              const int64_t row = row0 + r;
              const float t0 = s[2 * bp] + __shfl_xor(s[2 * bp], 32, 64);
              const float t1 = s[2 * bp + 1] + __shfl_xor(s[2 * bp + 1], 32, 64);
              float *dst = o_blk ? o_base + (int64_t)bp * o_blk + row * 64 + lane
                                 : o_base + row * (32 * NB) + 64 * bp + lane;
              *dst = (h ? t1 : t0) * f;      // lanes 0-31: columns 64 bp + c, lanes 32-63: 64 bp + 32 + c
              s[2 * bp] = 0.f;
              s[2 * bp + 1] = 0.f;
            }
          }
        }
      }
      d0 = d1; d1 = d2; d2 = d3;
      live = id1 >= 0;
      id1 = id2;
      ua = na;
      ub = nb;
      ux = nx;
    }
#undef H1_LOAD_DESC
#undef H1_LOAD_COL
#undef H1_LOAD_U
  }
}

// hub rows: the sum of the row's pieces, in piece order, times inv_deg * scale.  One wave per row.
__global__ void __launch_bounds__(256)
h1_hub_combine_k(const int4 *__restrict__ hubs, const int *__restrict__ hcnt,
                 const float *__restrict__ part, const float *__restrict__ inv_deg, float scale,
                 int Fo, float *__restrict__ out, int64_t out_blk) {
  const int lane = gs_lane();
  const int n_hubs = hcnt[1];
  const int nwaves = gridDim.x * 4;
  for (int hb = blockIdx.x * 4 + (threadIdx.x >> 6); hb < n_hubs; hb += nwaves) {
    const int4 e = hubs[hb];
    const int64_t row = e.x;
    const float f = inv_deg[row] * scale;
    for (int cb = 0; cb < Fo; cb += 64) {
      float acc = 0.f;
      const float *src = part + (int64_t)e.y * Fo + cb + lane;
      for (int k = 0; k < e.z; ++k) acc += src[(int64_t)k * Fo];
      float *dst = out_blk ? out + (int64_t)(cb >> 6) * out_blk + row * 64 + lane : out + row * Fo + cb + lane;
      *dst = acc * f;
    }
  }
}

constexpr size_t h1_planes_bytes(int64_t Fo) { return (size_t)(2 * Fo * 16) * sizeof(uint16_t); }

}  // namespace

extern "C" size_t gsage_hidden1_groups_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0) return 0;
  return h1_carve(nullptr, N, E).bytes;
}

extern "C" int gsage_hidden1_groups_build(const int32_t *rowptr, int64_t N, int64_t E, void *groups,
                                          size_t groups_bytes, gsage_stream_t stream) {
  if (N < 0 || E < 0 || !rowptr || !groups) return GSAGE_ERR_INVALID_ARG;
  if (N >= ((int64_t)1 << 26)) return GSAGE_ERR_UNSUPPORTED;      // 64-byte rows addressed by 32-bit offsets
  if (groups_bytes < gsage_hidden1_groups_bytes(N, E)) return GSAGE_ERR_WORKSPACE;
  if (N == 0) return GSAGE_OK;
  hipStream_t st = gs_stream(stream);
  const H1Groups g = h1_carve(groups, N, E);
  const int64_t nch = h1_chunks(N);
  hipLaunchKernelGGL(h1_groups_count_k, dim3((unsigned)((nch + 1 + 255) / 256)), dim3(256), 0, st, rowptr,
                     N, nch, g.goff);
  GSAGE_CHECK_LAUNCH();
  if (int rc = gs_exclusive_scan(g.goff, nch + 1, g.goff, g.bsum, st)) return rc;
  hipLaunchKernelGGL(h1_groups_fill_k, dim3((unsigned)((nch + 255) / 256)), dim3(256), 0, st, rowptr, N,
                     nch, (const int *)g.goff, g.gdesc);
  GSAGE_CHECK_LAUNCH();
  if (!gs_zero_async(g.hcnt, 4 * sizeof(int), st)) return GSAGE_ERR_LAUNCH;
  hipLaunchKernelGGL(h1_hub_list_k, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, rowptr, N, g.hcnt,
                     g.hubs, g.hub_cap, g.pieces, g.piece_cap);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_segment_gather_hidden1_supported(int64_t Fo, int epilogue) {
  if (epilogue == GSAGE_EPI_RELU) return (Fo == 64 || Fo == 128 || Fo == 256) ? 1 : 0;
  if (epilogue == GSAGE_EPI_RELU_DROPOUT) return (Fo == 64 || Fo == 128 || Fo == 256) ? 1 : 0;
  return 0;
}

// bytes per row of the table u[j]: 64, or 128 when the masks of more than 64 columns ride along
static inline size_t h1_urow_bytes(int64_t Fo, int epilogue) {
  return epilogue == GSAGE_EPI_RELU_DROPOUT && Fo > 64 ? 128 : 64;
}

extern "C" size_t gsage_segment_gather_hidden1_workspace_bytes(int64_t N, int64_t E, int64_t Fo, int epilogue) {
  if (N < 0 || E < 0 || Fo <= 0) return 0;
  // the rows u[j] and the word behind them, the weight planes, one partial row per piece of a hub
  // row, the work counter and the two scale factors
  return gs_align_up((size_t)(N + 1) * h1_urow_bytes(Fo, epilogue), 256) + gs_align_up(h1_planes_bytes(Fo), 256) +
         gs_align_up((size_t)h1_piece_cap(E) * (size_t)Fo * sizeof(float), 256) + 256;
}

// u_rows given: the table u[j] is the caller's (left by gsage_project_fwd_u, or by an earlier call),
// agg1 / x8 / h1 / relu_sign are not read and the workspace holds only planes, partial rows, counter
static int hidden1_impl(const void *groups, int64_t E, const int32_t *col, const float *inv_deg,
                        const float *agg1, const float *x8, const float *u_rows, const float *w_l,
                        const float *b_l, const float *w_r, int64_t N, int64_t Fo, int epilogue,
                        float p, const float *h1, const uint64_t *relu_sign, float *out,
                        int out_blocked, void *ws, size_t ws_bytes, gsage_stream_t stream) {
  if (N < 0 || E < 0 || !groups || !inv_deg || !w_l || !b_l || !w_r || !out || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (!u_rows && (!agg1 || !x8)) return GSAGE_ERR_INVALID_ARG;
  if (N >= ((int64_t)1 << 26)) return GSAGE_ERR_UNSUPPORTED;
  if (!gsage_segment_gather_hidden1_supported(Fo, epilogue)) return GSAGE_ERR_UNSUPPORTED;
  const bool train = epilogue == GSAGE_EPI_RELU_DROPOUT;
  if (train && !(p >= 0.f && p < 1.f)) return GSAGE_ERR_INVALID_ARG;
  if (train && !u_rows && !h1 && !relu_sign) return GSAGE_ERR_INVALID_ARG;
  if (out_blocked && Fo % 64 != 0) return GSAGE_ERR_UNSUPPORTED;
  if ((((uintptr_t)agg1 | (uintptr_t)x8 | (uintptr_t)h1 | (uintptr_t)ws | (uintptr_t)u_rows) & 15) != 0)
    return GSAGE_ERR_UNSUPPORTED;
  const size_t need = u_rows ? gsage_segment_gather_hidden1_u_workspace_bytes(E, Fo)
                             : gsage_segment_gather_hidden1_workspace_bytes(N, E, Fo, epilogue);
  if (ws_bytes < need) return GSAGE_ERR_WORKSPACE;
  // 32-bit byte offsets into the row table
  if ((uint64_t)N * h1_urow_bytes(Fo, epilogue) > ((uint64_t)1 << 32)) return GSAGE_ERR_UNSUPPORTED;
  if (N == 0) return GSAGE_OK;     // (col may be NULL: a graph without edges has nothing to index)
  hipStream_t st = gs_stream(stream);
  GsCarver carve(ws);
  const size_t urow = h1_urow_bytes(Fo, epilogue) / 4;      // floats per row
  const float *U = u_rows;
  if (!u_rows) U = carve.take<float>((size_t)(N + 1) * urow);
  // max |u| over the table as float bits: the first word of row N, behind the last row
  const uint32_t *u_amax = (const uint32_t *)(U + (size_t)N * urow);
  uint16_t *planes = carve.take<uint16_t>((size_t)(2 * Fo * 16));
  float *part = carve.take<float>((size_t)h1_piece_cap(E) * (size_t)Fo);
  int *counter = carve.take<int>(8);       // [0] the work counter, [4], [5] the two scale factors
  float *params = (float *)(counter + 4);
  if (!gs_zero_async(counter, 4 * sizeof(int), st)) return GSAGE_ERR_LAUNCH;
  if (!u_rows) {
    if (!gs_zero_async(const_cast<uint32_t *>(u_amax), sizeof(uint32_t), st)) return GSAGE_ERR_LAUNCH;
    hipLaunchKernelGGL(h1_pack_u_k, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, agg1, x8, N,
                       train ? 1 : 0, (int)(Fo / 64), h1, (const unsigned long long *)relu_sign,
                       const_cast<float *>(U), const_cast<uint32_t *>(u_amax));
  }
  hipLaunchKernelGGL(h1_pack_w_k, dim3(1), dim3(256), 0, st, w_l, b_l, w_r, (int)Fo, u_amax, planes, params);
  GSAGE_CHECK_LAUNCH();
  const unsigned grid = H1_GRID_WAVES / H1_WG_WAVES;
  const H1Groups g = h1_carve(const_cast<void *>(groups), N, E);
  const float scale = train ? 1.0f / (1.0f - p) : 1.0f;
  const int64_t blk = out_blocked ? N * 64 : 0;
#define H1_LAUNCH(NB, TR)                                                                        \
  hipLaunchKernelGGL((hidden1_gather_k<NB, TR>), dim3(grid), dim3(64 * H1_WG_WAVES), 0, st,                  \
                     (const int *)g.goff, (const int2 *)g.gdesc, col, inv_deg, (const float *)U, \
                     (const uint16_t *)planes, (const float *)params, N, scale, out, blk, (const int2 *)g.pieces, \
                     (const int *)g.hcnt, part, counter)
  if (train && Fo == 64) H1_LAUNCH(2, true);
  else if (train && Fo == 128) H1_LAUNCH(4, true);
  else if (train) H1_LAUNCH(8, true);
  else if (Fo == 64) H1_LAUNCH(2, false);
  else if (Fo == 128) H1_LAUNCH(4, false);
  else H1_LAUNCH(8, false);
#undef H1_LAUNCH
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(h1_hub_combine_k, dim3(256), dim3(256), 0, st, (const int4 *)g.hubs, (const int *)g.hcnt,
                     (const float *)part, inv_deg, scale, (int)Fo, out, blk);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_segment_gather_hidden1(const void *groups, int64_t E, const int32_t *col,
                                            const float *inv_deg, const float *agg1, const float *x8,
                                            const float *w_l, const float *b_l, const float *w_r,
                                            int64_t N, int64_t Fo, int epilogue, float p,
                                            const float *h1, const uint64_t *relu_sign, float *out,
                                            int out_blocked, void *ws, size_t ws_bytes,
                                            gsage_stream_t stream) {
  if (!agg1 || !x8) return GSAGE_ERR_INVALID_ARG;
  return hidden1_impl(groups, E, col, inv_deg, agg1, x8, nullptr, w_l, b_l, w_r, N, Fo, epilogue, p,
                      h1, relu_sign, out, out_blocked, ws, ws_bytes, stream);
}

extern "C" size_t gsage_hidden1_u_row_bytes(int64_t Fo, int epilogue) { return h1_urow_bytes(Fo, epilogue); }
// the table has N + 1 rows: the first word of row N receives max |u| over the table (float bits),
// which the gather scales its fp16 operands by
extern "C" int64_t gsage_hidden1_u_rows(int64_t N) { return N + 1; }

extern "C" size_t gsage_segment_gather_hidden1_u_workspace_bytes(int64_t E, int64_t Fo) {
  if (E < 0 || Fo <= 0) return 0;
  return gs_align_up(h1_planes_bytes(Fo), 256) +
         gs_align_up((size_t)h1_piece_cap(E) * (size_t)Fo * sizeof(float), 256) + 256;
}

extern "C" int gsage_segment_gather_hidden1_u(const void *groups, int64_t E, const int32_t *col,
                                              const float *inv_deg, const float *u_rows,
                                              const float *w_l, const float *b_l, const float *w_r,
                                              int64_t N, int64_t Fo, int epilogue, float p, float *out,
                                              int out_blocked, void *ws, size_t ws_bytes,
                                              gsage_stream_t stream) {
  if (!u_rows) return GSAGE_ERR_INVALID_ARG;
  return hidden1_impl(groups, E, col, inv_deg, nullptr, nullptr, u_rows, w_l, b_l, w_r, N, Fo,
                      epilogue, p, nullptr, nullptr, out, out_blocked, ws, ws_bytes, stream);
}
