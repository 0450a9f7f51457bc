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
// One wave per chunk of 64 destination rows; their neighbours are walked in groups of 32 (four
// slots of 8, see the group index below): the 32 rows u[j] are the A operand of
// v_mfma_f32_32x32x16_bf16 (K = 16 is exactly the layer's input width, the bias rides in the
// padding column 7), every fp32 operand as the exact sum of three bf16 parts and the six largest
// of the nine products issued, smallest first (project.hip, gemm_split_k: fp32-equivalent
// accuracy).  The accumulator comes out as D[neighbour][column] with the neighbours in the
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

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

#ifndef H1_ROWS_PER_WAVE
#define H1_ROWS_PER_WAVE 64
#endif
constexpr int H1_ROWS = H1_ROWS_PER_WAVE;   // destination rows per wave, at most 64 (one trip per wave, in-order dispatch)

// ---- weight planes: Wc[n][k] = k < 7 ? W_l[n][k] : k == 7 ? b[n] : k < 15 ? W_r[n][k-8] : 0,
// split into three bf16 planes [3][Fo][16]
__global__ void __launch_bounds__(256) h1_pack_w_k(const float *__restrict__ wl, const float *__restrict__ bl,
                                                   const float *__restrict__ wr, int Fo,
                                                   uint16_t *__restrict__ planes) {
  const int total = Fo * 16;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int n = e >> 4, k = e & 15;
    const float v = k < 7 ? wl[n * 8 + k] : k == 7 ? bl[n] : k < 15 ? wr[n * 8 + (k - 8)] : 0.f;
    const uint32_t u1 = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(u1);
    const uint32_t u2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(u2);
    planes[e] = (uint16_t)(u1 >> 16);
    planes[total + e] = (uint16_t)(u2 >> 16);
    planes[2 * total + e] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

// ---- the 64-byte rows: u[j] = [agg1[j][0..6], m_lo, x[j][0..6], m_hi].  m = the 64 keep-and-positive
// bits of row j (Fo = 64, training), bit 16 cc + q of the 64-bit mask = column 4 q + cc: the order
// the first-layer kernel's ballots come in (gsage_relu_sign_bytes: per 4 rows and column block four
// 64-bit words, word cc bit l = column 4 (l & 15) + cc of row l >> 4), so a row's mask is the four
// 16-bit fields of its group's words.  No sign bits (other forward kernels): taken from h1 > 0.
__global__ void __launch_bounds__(256) h1_pack_u_k(const float *__restrict__ agg1, const float *__restrict__ x8,
                                                   int64_t N, int train, const float *__restrict__ h1,
                                                   const unsigned long long *__restrict__ sign,
                                                   float *__restrict__ U) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // the flag word behind the bits says whether the forward kernel that ran wrote them
  const bool have_bits = train && sign && sign[((N + 3) >> 2) * 4] != 0ull;
  if (j >= N) return;
  f32x4v a0 = *(const f32x4v *)(agg1 + j * 8), a1 = *(const f32x4v *)(agg1 + j * 8 + 4);
  f32x4v b0 = *(const f32x4v *)(x8 + j * 8), b1 = *(const f32x4v *)(x8 + j * 8 + 4);
  uint32_t lo = 0u, hi = 0u;
  if (train) {
    if (have_bits) {
      const unsigned long long *w = sign + (j >> 2) * 4;
      const int sh = 16 * (int)(j & 3);
      lo = (uint32_t)((w[0] >> sh) & 0xFFFFull) | ((uint32_t)((w[1] >> sh) & 0xFFFFull) << 16);
      hi = (uint32_t)((w[2] >> sh) & 0xFFFFull) | ((uint32_t)((w[3] >> sh) & 0xFFFFull) << 16);
    } else {
      const float *r = h1 + j * 64;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4v v = *(const f32x4v *)(r + 4 * q);
        lo |= (v[0] > 0.f ? 1u : 0u) << q | (v[1] > 0.f ? 1u : 0u) << (16 + q);
        hi |= (v[2] > 0.f ? 1u : 0u) << q | (v[3] > 0.f ? 1u : 0u) << (16 + q);
      }
    }
  }
  a1[3] = __uint_as_float(lo);
  b1[3] = __uint_as_float(hi);
  float *u = U + j * 16;
  *(f32x4v *)(u) = a0;
  *(f32x4v *)(u + 4) = a1;
  *(f32x4v *)(u + 8) = b0;
  *(f32x4v *)(u + 12) = b1;
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
constexpr int H1_SLOT = 8;
constexpr int H1_META_LAST = 1 << 10, H1_META_VALID = 1 << 11;

__device__ __forceinline__ int h1_row_slots(int deg) { return deg > 0 ? (deg + H1_SLOT - 1) / H1_SLOT : 1; }

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
  return g;
}

template <int NB, bool TRAIN>
__global__ void __launch_bounds__(256)
hidden1_gather_k(const int *__restrict__ goff, const int2 *__restrict__ gdesc,
                 const int *__restrict__ col, const float *__restrict__ inv_deg,
                 const float *__restrict__ U, const uint16_t *__restrict__ Wp, int64_t N, float scale,
                 float *__restrict__ out, int64_t out_blk) {
  static_assert(!TRAIN || NB == 2, "the 64 mask bits of a row cover 64 columns");
  const int lane = gs_lane(), c = lane & 31, h = lane >> 5;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row0 = chunk * H1_ROWS;
  if (row0 >= N) return;
  const int g0 = goff[chunk], g1 = goff[chunk + 1];

  // B operand: column 32 b + c, k = 8 h .. 8 h + 7 of the three weight planes, for the whole launch
  bf16x8 wf[NB][3];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      wf[b][pl] = *(const bf16x8 *)(Wp + ((int64_t)pl * 32 * NB + 32 * b + c) * 16 + 8 * h);
  const float idg = row0 + lane < N && lane < H1_ROWS ? inv_deg[row0 + lane] : 0.f;   // lane k: row k's scale

  float s[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) s[b] = 0.f;
  // training: which mask word a lane reads (columns with c & 2: the high word, held by the lanes
  // of the upper half) and which bit of it per column block
  const int bp_base = 4 * (((c >> 1) & 1) * 32 + 4 * h);
  const int bit0 = (c & 1) * 16 + (c >> 2);
  const uint32_t u_lane = 32u * (uint32_t)h;       // byte offset of this lane's half of a 64-byte row
  const int sl = c >> 3, ln = c & 7;               // this lane's slot of a group, its place in the slot

  // Three loads hang on one another -- slot descriptor, neighbour id, the row u[j] -- so the loop
  // runs them as a pipeline: while group g is computed, the rows of g + 1, the ids of g + 2 and
  // the descriptors of g + 3 are in flight.
#define H1_LOAD_DESC(DST, G)                                                                  \
  {                                                                                           \
    DST = make_int2(0, 0);                                                                    \
    if ((G) < g1) DST = gdesc[(int64_t)(G) * 4 + sl];                                         \
  }
#define H1_LOAD_COL(DST, DESC)                                                                \
  {                                                                                           \
    DST = -1;                                                                                 \
    if (ln < ((DESC).y & 15)) DST = col[(DESC).x + ln];                                       \
  }
#define H1_LOAD_U(A, B, ID)                                                                   \
  {                                                                                           \
    A = (f32x4v){0.f, 0.f, 0.f, 0.f};                                                         \
    B = (f32x4v){0.f, 0.f, 0.f, 0.f};                                                         \
    if ((ID) >= 0) {                                                                          \
      const char *p_ = (const char *)U + ((uint32_t)(ID) * 64u + u_lane);                     \
      A = *(const f32x4v *)p_;                                                                \
      B = *(const f32x4v *)(p_ + 16);                                                         \
    }                                                                                         \
  }
  int2 d0, d1, d2, d3;          // descriptors of group g, g + 1, g + 2, g + 3 (this lane's slot)
  int id1, id2;                 // neighbour ids of g + 1, g + 2
  f32x4v ua, ub;                // the row of g
  bool live;
  {
    int id0;
    H1_LOAD_DESC(d0, g0)
    H1_LOAD_DESC(d1, g0 + 1)
    H1_LOAD_DESC(d2, g0 + 2)
    H1_LOAD_COL(id0, d0)
    H1_LOAD_COL(id1, d1)
    H1_LOAD_U(ua, ub, id0)
    live = id0 >= 0;
  }
  for (int g = g0; g < g1; ++g) {
    f32x4v na, nb;
    H1_LOAD_U(na, nb, id1)
    H1_LOAD_COL(id2, d2)
    H1_LOAD_DESC(d3, g + 3)
    // ---- this group: A operand = row of neighbour c, k = 8 h .. 8 h + 7
    uint32_t mword = 0u;
    if (TRAIN) mword = live ? __float_as_uint(ub[3]) : 0u;
    ub[3] = (live && h == 0) ? 1.f : 0.f;          // k = 7: the bias column; k = 15: nothing
    uint32_t p1[8], p2[8], p3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = e < 4 ? ua[e] : ub[e - 4];
      p1[e] = __float_as_uint(v) & 0xFFFF0000u;
      const float r1 = v - __uint_as_float(p1[e]);
      p2[e] = __float_as_uint(r1) & 0xFFFF0000u;
      p3[e] = __float_as_uint(r1 - __uint_as_float(p2[e]));
    }
    // two high halves into one word: bytes 3:2 of the first operand above those of the second
#define H1_PK(P, E) __builtin_amdgcn_perm(P[(E) + 1], P[E], 0x07060302u)
    const u32x4v w1 = {H1_PK(p1, 0), H1_PK(p1, 2), H1_PK(p1, 4), H1_PK(p1, 6)};
    const u32x4v w2 = {H1_PK(p2, 0), H1_PK(p2, 2), H1_PK(p2, 4), H1_PK(p2, 6)};
    const u32x4v w3 = {H1_PK(p3, 0), H1_PK(p3, 2), H1_PK(p3, 4), H1_PK(p3, 6)};
#undef H1_PK
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, w1), a2 = __builtin_bit_cast(bf16x8, w2),
                 a3 = __builtin_bit_cast(bf16x8, w3);
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
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, wf[b][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[b][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wf[b][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, wf[b][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[b][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wf[b][0], acc, 0, 0, 0);
        // acc[r] = pre-activation of neighbour (r & 3) + 8 (r >> 2) + 4 h, column 32 b + c
        float v[16];
        if (TRAIN) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int nb_ = (r & 3) + 8 * (r >> 2);
            const uint32_t m = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + 4 * nb_, (int)mword);
            const int keep = __builtin_amdgcn_sbfe((int)m, bit0 + 8 * b, 1);     // 0 or -1
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
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int meta = __builtin_amdgcn_readlane(d0.y, 8 * q);
        if (meta & H1_META_VALID) {
          s[2 * bp] += t[0][q];
          s[2 * bp + 1] += t[1][q];
          if (meta & H1_META_LAST) {
            const int r = (meta >> 4) & 63;
            const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, idg), r)) * scale;
            const int64_t row = row0 + r;
            const float t0 = s[2 * bp] + __shfl_xor(s[2 * bp], 32, 64);
            const float t1 = s[2 * bp + 1] + __shfl_xor(s[2 * bp + 1], 32, 64);
            float *dst = out_blk ? out + (int64_t)bp * out_blk + row * 64 + lane
                                 : out + row * (32 * NB) + 64 * bp + lane;
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
  }
#undef H1_LOAD_DESC
#undef H1_LOAD_COL
#undef H1_LOAD_U
}

// ---- K7h: the input layer's weight gradients straight from the layer-2 backward, without the
// dense gradient of the hidden activation in between.  PyG's backward (graphsage.py:133-138 under
// autograd) first scatters d(loss)/d(agg2) back over the edges into d(loss)/d(h1) [N, 64] and then
// contracts that with the input layer's operands.  With g_i = inv_deg[i] * d(loss)/d(agg2)[i]
// (non-zero on the rows i the batch touched), r_j = the root-path gradient of row j and m_j the
// keep-and-positive bits of h1[j] (the bits in u[j], see above):
//   dW[c][k] = sum_{live i} sum_{j in N(i)} m_j[c] g_i[c] u_j[k]  +  sum_{live j} m_j[c] r_j[c] u_j[k]
// with u_j = [agg1[j] | 1 | x[j] | 0], so column 7 of dW is the bias gradient.  Every live edge
// (i <- j) costs the 64-byte row u[j] (one line) where the transposed gather fetches the
// 256-byte row g_i per edge of j (two lines), and nothing of size N x 64 is written or re-read.
// v_mfma_f32_16x16x4_f32 takes its K dimension from the lane GROUP (lane >> 4): four neighbours
// sit in the four 16-lane groups of a wave, lane (group n, f) holds u[n][f] as the B operand
// and m_n[c] g_i[c] for c = 16 t + f as the A operand of tile t -- exact fp32 products, no
// transposes, no operand splitting; the 64 x 16 result stays in 16 accumulator registers per
// lane for the whole launch.
constexpr int H1B_GRID = 256 * 8;                 // persistent workgroups
constexpr int H1B_SLAB = 64 * 16;                 // dW[c][k] of one workgroup

__global__ void __launch_bounds__(256)
hidden1_dweights_k(const int *__restrict__ rows, const int *__restrict__ n_rows, int rows_cap,
                   const int *__restrict__ rowptr, const int *__restrict__ col,
                   const float *__restrict__ U, const float *__restrict__ gagg,
                   const float *__restrict__ groot, float inv_keep, float *__restrict__ slabs) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[4][16][64];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  const int f = lane & 15, ng = lane >> 4;
  const int n = *n_rows < rows_cap ? *n_rows : rows_cap;
  const int bit0 = (f & 1) * 16 + (f >> 2);       // bit of column 16 t + f in its mask word: bit0 + 4 t
  const int lo_lane = 4 * ((lane & 48) | 7), hi_lane = 4 * ((lane & 48) | 15);   // bpermute addresses
  // The matrix pipe's accumulation is not what sums the launch: chains of MFMAs into one
  // accumulator showed a rounding bias that grows with the chain (same-sign terms, as the bias
  // gradient of a hub row has them: 4e-6 of the largest gradient in the parity test, seven times
  // the error of the weight columns).  So a chain is at most 36 neighbours long: the accumulators
  // are added to the running total `tot` (vector adds, round to nearest) every four steps of a
  // row and at its end.
  f32x4 acc[1][4], tot[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    tot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#define H1B_FLUSH()                                                                           \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                             \
    tot[t] += acc[0][t];                                                                      \
    acc[0][t] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                  \
  }

  // one quad: lane group ng holds neighbour row `id` (-1: none) and the gradient vector `vec`
#define H1B_QUAD(SET, UVAL, VALID, VEC)                                                       \
  {                                                                                           \
    const int ub_ = __float_as_int(UVAL);                                                     \
    const int mlo_ = __builtin_amdgcn_ds_bpermute(lo_lane, ub_);                              \
    const int mhi_ = __builtin_amdgcn_ds_bpermute(hi_lane, ub_);                              \
    const int w_ = (f & 2) ? mhi_ : mlo_;                                                     \
    const float b_ = f == 7 ? ((VALID) ? 1.f : 0.f) : (f == 15 ? 0.f : (UVAL));               \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                           \
      const int keep_ = __builtin_amdgcn_sbfe(w_, bit0 + 4 * t, 1);                           \
      const float a_ = __int_as_float(keep_ & __float_as_int((VEC)[t]));                      \
      acc[SET][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, acc[SET][t], 0, 0, 0);       \
    }                                                                                         \
  }
#ifdef H1B_CAP
#define H1B_CAP_STMT if (p1 > p0 + H1B_CAP) p1 = p0 + H1B_CAP;
#else
#define H1B_CAP_STMT
#endif
#define H1B_ROW(SET, R)                                                                       \
  {                                                                                           \
    const int i = __builtin_amdgcn_readlane(iv, (R));                                         \
    const int p0 = __builtin_amdgcn_readlane(p0v, (R));                                       \
    int p1 = __builtin_amdgcn_readlane(p1v, (R));                                             \
    H1B_CAP_STMT                                                                              \
    /* neighbour ids and rows of the first two quads, the row's own operands */              \
    int ida = p0 + ng < p1 ? col[p0 + ng] : -1;                                               \
    int idb = p0 + 4 + ng < p1 ? col[p0 + 4 + ng] : -1;                                       \
    float vg[4], vr[4];                                                                       \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                           \
      vg[t] = gagg[(int64_t)i * 64 + 16 * t + f];                                             \
      vr[t] = groot[(int64_t)i * 64 + 16 * t + f];                                            \
    }                                                                                         \
    const float uself = ng == 0 ? U[(int64_t)i * 16 + f] : 0.f;                               \
    float ua = 0.f, ub = 0.f;                                                                 \
    if (ida >= 0) ua = U[(int64_t)ida * 16 + f];                                              \
    if (idb >= 0) ub = U[(int64_t)idb * 16 + f];                                              \
    /* the root path: the row itself as the only neighbour of a quad, with its own vector */  \
    H1B_QUAD(SET, uself, ng == 0, vr)                                                         \
    int step_ = 0;                                                                            \
    for (int p = p0; p < p1; p += 8, ++step_) {                                               \
      if ((step_ & 3) == 3) H1B_FLUSH()                                                       \
      /* one step ahead: ids, then rows */                                                    \
      const int q = p + 8 + ng;                                                               \
      const int nida = q < p1 ? col[q] : -1;                                                  \
      const int nidb = q + 4 < p1 ? col[q + 4] : -1;                                          \
      const bool va = ida >= 0, vb = idb >= 0;                                                \
      const float ca = ua, cb = ub;                                                           \
      ua = 0.f; ub = 0.f;                                                                     \
      if (nida >= 0) ua = U[(int64_t)nida * 16 + f];                                          \
      if (nidb >= 0) ub = U[(int64_t)nidb * 16 + f];                                          \
      ida = nida; idb = nidb;                                                                 \
      H1B_QUAD(SET, ca, va, vg)                                                               \
      H1B_QUAD(SET, cb, vb, vg)                                                               \
    }                                                                                         \
    H1B_FLUSH()                                                                               \
  }

  const int gwave = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  for (int base = gwave * 64; base < n; base += nwaves * 64) {
    // this wave's next 64 live rows, one per lane
    const int nb = n - base < 64 ? n - base : 64;
    int iv = 0, p0v = 0, p1v = 0;
    if (lane < nb) {
      iv = rows[base + lane];
      p0v = rowptr[iv];
      p1v = rowptr[iv + 1];
    }
    for (int r = 0; r < nb; ++r) H1B_ROW(0, r)
  }
#undef H1B_ROW
#undef H1B_QUAD
#undef H1B_FLUSH
  // acc[t][r] of lane l = dW[c = 16 t + 4 (l >> 4) + r][k = l & 15]; the workgroup's four waves
  // are summed in wave order, the workgroups by h1_dweights_reduce_k in workgroup order
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][4 * t + r][lane] = tot[t][r] * inv_keep;
  __syncthreads();
  float *slab = slabs + (int64_t)blockIdx.x * H1B_SLAB;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int q = e >> 6, l = e & 63;
    const float v = ((red[0][q][l] + red[1][q][l]) + red[2][q][l]) + red[3][q][l];
    const int c = 16 * (q >> 2) + 4 * (l >> 4) + (q & 3);
    slab[c * 16 + (l & 15)] = v;
  }
}

__global__ void __launch_bounds__(256)
h1_dweights_reduce_k(const float *__restrict__ slabs, int nslab, float *__restrict__ gwl,
                     float *__restrict__ gb, float *__restrict__ gwr) {
  __shared__ float red[4][64];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;           // element c * 16 + k of the 64 x 16 result
  float s = 0.f;
  for (int b = wave; b < nslab; b += 4) s += slabs[(int64_t)b * H1B_SLAB + e];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const float v = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int c = e >> 4, k = e & 15;
    if (k < 7) gwl[c * 8 + k] = v;
    else if (k == 7) { gb[c] = v; gwl[c * 8 + 7] = 0.f; }      // the padding column's gradient is zero
    else if (k < 15) gwr[c * 8 + (k - 8)] = v;
    else gwr[c * 8 + 7] = 0.f;
  }
}

constexpr size_t h1_planes_bytes(int64_t Fo) { return (size_t)(3 * Fo * 16) * sizeof(uint16_t); }

}  // namespace

extern "C" size_t gsage_hidden1_groups_bytes(int64_t N, int64_t E) {
  if (N < 0 || E < 0) return 0;
  const int64_t nch = h1_chunks(N);
  return gs_align_up((size_t)(nch + 1) * sizeof(int), 256) +
         gs_align_up((size_t)((nch + 1 + GS_SCAN_CH - 1) / GS_SCAN_CH + 1) * sizeof(int), 256) +
         gs_align_up((size_t)(4 * h1_max_groups(N, E)) * sizeof(int2), 256);
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
  return GSAGE_OK;
}

extern "C" int gsage_segment_gather_hidden1_supported(int64_t Fo, int epilogue) {
  if (epilogue == GSAGE_EPI_RELU) return (Fo == 64 || Fo == 128 || Fo == 256) ? 1 : 0;
  if (epilogue == GSAGE_EPI_RELU_DROPOUT) return Fo == 64 ? 1 : 0;
  return 0;
}

extern "C" size_t gsage_segment_gather_hidden1_workspace_bytes(int64_t N, int64_t Fo) {
  if (N < 0 || Fo <= 0) return 0;
  return gs_align_up((size_t)N * 16 * sizeof(float), 256) + gs_align_up(h1_planes_bytes(Fo), 256);
}

extern "C" int gsage_segment_gather_hidden1(const void *groups, int64_t E, const int32_t *col,
                                            const float *inv_deg, const float *agg1, const float *x8,
                                            const float *w_l, const float *b_l, const float *w_r,
                                            int64_t N, int64_t Fo, int epilogue, float p,
                                            const float *h1, const uint64_t *relu_sign, float *out,
                                            int out_blocked, void *ws, size_t ws_bytes,
                                            gsage_stream_t stream) {
  if (N < 0 || E < 0 || !groups || !inv_deg || !agg1 || !x8 || !w_l || !b_l || !w_r || !out || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (N >= ((int64_t)1 << 26)) return GSAGE_ERR_UNSUPPORTED;
  if (!gsage_segment_gather_hidden1_supported(Fo, epilogue)) return GSAGE_ERR_UNSUPPORTED;
  const bool train = epilogue == GSAGE_EPI_RELU_DROPOUT;
  if (train && (!(p >= 0.f && p < 1.f) || (!h1 && !relu_sign))) return GSAGE_ERR_INVALID_ARG;
  if (out_blocked && Fo % 64 != 0) return GSAGE_ERR_UNSUPPORTED;
  if ((((uintptr_t)agg1 | (uintptr_t)x8 | (uintptr_t)h1 | (uintptr_t)ws) & 15) != 0) return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < gsage_segment_gather_hidden1_workspace_bytes(N, Fo)) return GSAGE_ERR_WORKSPACE;
  if (N == 0) return GSAGE_OK;     // (col may be NULL: a graph without edges has nothing to index)
  hipStream_t st = gs_stream(stream);
  GsCarver carve(ws);
  float *U = carve.take<float>((size_t)N * 16);
  uint16_t *planes = carve.take<uint16_t>((size_t)(3 * Fo * 16));
  hipLaunchKernelGGL(h1_pack_w_k, dim3(gs_grid(Fo * 16, 256, 16)), dim3(256), 0, st, w_l, b_l, w_r,
                     (int)Fo, planes);
  hipLaunchKernelGGL(h1_pack_u_k, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, agg1, x8, N,
                     train ? 1 : 0, h1, (const unsigned long long *)relu_sign, U);
  GSAGE_CHECK_LAUNCH();
  const unsigned grid = (unsigned)((N + 4 * H1_ROWS - 1) / (4 * H1_ROWS));
  const H1Groups g = h1_carve(const_cast<void *>(groups), N, E);
  const float scale = train ? 1.0f / (1.0f - p) : 1.0f;
  const int64_t blk = out_blocked ? N * 64 : 0;
#define H1_LAUNCH(NB, TR)                                                                        \
  hipLaunchKernelGGL((hidden1_gather_k<NB, TR>), dim3(grid), dim3(256), 0, st,                  \
                     (const int *)g.goff, (const int2 *)g.gdesc, col, inv_deg, (const float *)U, \
                     (const uint16_t *)planes, N, scale, out, blk)
  if (train) H1_LAUNCH(2, true);
  else if (Fo == 64) H1_LAUNCH(2, false);
  else if (Fo == 128) H1_LAUNCH(4, false);
  else H1_LAUNCH(8, false);
#undef H1_LAUNCH
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" size_t gsage_hidden1_dweights_workspace_bytes(void) {
  return (size_t)H1B_GRID * H1B_SLAB * sizeof(float);
}

extern "C" int gsage_hidden1_dweights(const int32_t *live_rows, const int32_t *n_live, int64_t live_cap,
                                      const int32_t *rowptr, const int32_t *col, const float *u_rows,
                                      const float *grad_agg, const float *grad_root, int64_t N, float p,
                                      float *grad_w_l, float *grad_b, float *grad_w_r, void *ws,
                                      size_t ws_bytes, gsage_stream_t stream) {
  if (N < 0 || live_cap < 0 || !live_rows || !n_live || !rowptr || !u_rows || !grad_agg || !grad_root ||
      !grad_w_l || !grad_b || !grad_w_r || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (!(p >= 0.f && p < 1.f)) return GSAGE_ERR_INVALID_ARG;
  if (N >= ((int64_t)1 << 26) || live_cap >= ((int64_t)1 << 31)) return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < gsage_hidden1_dweights_workspace_bytes()) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  float *slabs = (float *)ws;
  // a fixed grid whatever the batch: every workgroup writes its slab (zeros when it had no rows),
  // so the sum below has the same terms in the same order on every call
  hipLaunchKernelGGL(hidden1_dweights_k, dim3(H1B_GRID), dim3(256), 0, st, live_rows, n_live,
                     (int)live_cap, rowptr, col, u_rows, grad_agg, grad_root, 1.0f / (1.0f - p), slabs);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(h1_dweights_reduce_k, dim3(H1B_SLAB / 64), dim3(256), 0, st, (const float *)slabs,
                     H1B_GRID, grad_w_l, grad_b, grad_w_r);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

