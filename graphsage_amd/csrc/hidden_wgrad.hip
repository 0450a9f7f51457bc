// K7h: weight gradients of the INPUT layer, taken straight from the backward of the layer behind it.
//
// The reference's backward walks graphsage.py:133-138 in reverse: the gradient of layer 2's
// aggregation, ga[i] (one 64-float row per destination i), is scattered to the neighbours j of i,
// the sum gh[j] = sum_{i : j in N(i)} inv_deg[i] ga[i] + gx[j] goes through ReLU'/dropout' of
// layer 1 and only then meets layer 1's operands,
//
//   dW1[c][k] = sum_j m[j][c] gh[j][c] uu[j][k] / (1 - p),     uu[j] = [agg1[j] | 1 | x[j] | 0],
//
// (k < 7: W_l, k = 7: the bias, 8 <= k < 15: W_r) -- nothing else is wanted from layer 1, its input
// is the feature matrix.  The transposed gather that forms gh fetches 256 bytes (two fabric lines)
// per live edge and writes an N x 64 table that one more pass reads again.  Summed the other way
// round,
//
//   dW1[c][k] = sum_{i touched} sum_{j in N(i)} (inv_deg[i] ga[i][c]) m[j][c] uu[j][k]  +  the same
//               with gx[i] and j = i,
//
// every live edge fetches the 64-byte row u[j] = [agg1[j] | x[j]] with the 64 mask bits of j in its
// two padding words (the table the forward's recompute gather reads, hidden_gather.hip) -- ONE line
// per edge, nothing written -- and ga[i] once per 8 neighbours.  It is one long GEMM over the edges:
// K = live edges, M = 16 (the columns of uu), N = 64.
//
// A wave takes 32 neighbours at a time (v_mfma_f32_16x16x32_f16: K = 32): four SLOTS of 8
// neighbours of one destination each, so the B operand of a slot is that destination's gradient row
// with the neighbours' mask bits applied.  The slots of the touched rows are listed once per step
// from the row list the loss backward leaves; since nothing is stored per row, slots of any rows
// share an instruction and hub rows need no special care.  The ROOT terms (j = i against gx[i])
// follow in a second phase of the same launch, 32 listed rows per instruction with one gradient
// row per neighbour (as slots of their own, one "neighbour" each, they were 17 % of all slots and
// 0.42 of the kernel's 3.44 ms).  fp32 operands go to the fp16 pipe as in the forward: two fp16 parts each, scaled by
// powers of two from max |u| (behind the table) and max |g| (published by the kernel that writes
// ga / gx), three of the four products.
// The launch is a fixed number of waves, each owning every WG_WAVES-th block of 16 groups and
// leaving one 16 x 64 partial; two small kernels add the partials in a fixed order: the result does
// not depend on scheduling (bitwise reproducible).
#include "gsage_common.h"
#include "gsage_sort.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

constexpr float WG_TOP = 8192.f;         // 2^13: where the largest operand magnitudes are put
#ifndef WG_WAVES_
#define WG_WAVES_ 6144
#endif
#ifndef WG_BLOCK_
#define WG_BLOCK_ 16
#endif
constexpr int WG_WAVES = WG_WAVES_;      // waves of the launch (one per workgroup): 256 CUs x 4 SIMDs x 6 (80 registers)
constexpr int WG_BLOCK = WG_BLOCK_;      // groups per block (64 slot descriptors = 512 contiguous bytes); a power of two
#ifndef WG_FLUSH_
#define WG_FLUSH_ 8
#endif
constexpr int WG_FLUSH = WG_FLUSH_;      // groups per accumulation chain (a power of two)
constexpr int WG_SLICES = 64;            // first reduction: WG_WAVES / WG_SLICES partials each
// slot descriptor: (first position in col, count (0..8) | row << 5)

__device__ __forceinline__ float wg_pow2_floor(float v, float fallback) {
  const uint32_t e = (__float_as_uint(v) >> 23) & 255u;
  if (!(v > 0.f) || e < 27u || e > 227u) return fallback;
  return __uint_as_float(e << 23);
}

// params: [0] rows -> fp16 range, [1] gradients -> fp16 range, [2] undoes both and applies 1 / (1 - p),
// [3] the same for the bias column (whose operand is the constant 2^13)
__global__ void wg_params_k(const uint32_t *__restrict__ u_amax, const uint32_t *__restrict__ g_amax,
                            float inv_keep, float *__restrict__ params) {
  const float Ap = wg_pow2_floor(__uint_as_float(*u_amax), 1.f);
  const float Gp = wg_pow2_floor(__uint_as_float(*g_amax), 1.f);
  const float su = WG_TOP / Ap, sg = WG_TOP / Gp;
  params[0] = su;
  params[1] = sg;
  params[2] = inv_keep / (su * sg);
  params[3] = inv_keep / (WG_TOP * sg);
}

// slots per listed row: ceil(deg / 8); counts[cap] = 0 (the scan leaves the total there)
__global__ void __launch_bounds__(256) wg_count_k(const int *__restrict__ rows, const int *__restrict__ n_rows,
                                                  int64_t cap, const int *__restrict__ rowptr,
                                                  int *__restrict__ counts) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t > cap) return;
  int64_t n = *n_rows;
  if (n > cap) n = cap;
  int c = 0;
  if (t < n) {
    const int r = rows[t];
    c = (rowptr[r + 1] - rowptr[r] + 7) / 8;
  }
  counts[t] = c;
}

__global__ void __launch_bounds__(256) wg_fill_k(const int *__restrict__ rows, const int *__restrict__ n_rows,
                                                 int64_t cap, const int *__restrict__ rowptr,
                                                 const int *__restrict__ soff, int2 *__restrict__ desc) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = gs_lane();
  int64_t n = *n_rows;
  if (n > cap) n = cap;
  const bool has = t < n;
  int r = 0, p0 = 0, deg = 0, base = 0;
  if (has) {
    r = rows[t];
    p0 = rowptr[r];
    deg = rowptr[r + 1] - p0;
    base = soff[t];
  }
  const int ns = (deg + 7) / 8;
  const bool longrow = ns > 16;
  if (has && !longrow)
    for (int q = 0; q < ns; ++q) {
      const int left = deg - 8 * q;
      desc[base + q] = make_int2(p0 + 8 * q, (left < 8 ? left : 8) | (r << 5));
    }
  // the long rows of a wave: all its lanes write the slots
  unsigned long long m = __ballot(longrow);
  while (m) {
    const int l = __builtin_ctzll(m);
    m &= m - 1;
    const int r_ = __shfl(r, l, 64), p_ = __shfl(p0, l, 64), d_ = __shfl(deg, l, 64), b_ = __shfl(base, l, 64);
    const int ns_ = (d_ + 7) / 8;
    for (int q = lane; q < ns_; q += 64) {
      const int left = d_ - 8 * q;
      desc[b_ + q] = make_int2(p_ + 8 * q, (left < 8 ? left : 8) | (r_ << 5));
    }
  }
}

// lane = (n = lane & 15, kg = lane >> 4).  As the A operand's lane: column n of uu (M), slot kg's 8
// neighbours (K = 8 kg + e).  As the B operand's lane: gradient column 16 b + n (N) of slot kg.
// D[m][col]: lane holds col = 16 b + n, rows m = 4 kg + r in register r of accumulator b.
#ifndef WG_OCC_
#define WG_OCC_ 6
#endif
__global__ void __launch_bounds__(64, WG_OCC_)
hidden1_wgrad_k(const int2 *__restrict__ desc, const int *__restrict__ n_slots, const int *__restrict__ col,
                const float *__restrict__ U, const float *__restrict__ ga, const float *__restrict__ gx,
                const float *__restrict__ params, float *__restrict__ partial,
                const int *__restrict__ rows, const int *__restrict__ n_rows, int rows_cap) {
  const int lane = threadIdx.x, n = lane & 15, kg = lane >> 4;
  const int S = *n_slots;
  const int G = (S + 3) >> 2;
  const int nblocks = (G + WG_BLOCK - 1) / WG_BLOCK;
  const float su = params[0], sg = params[1];
  // The matrix pipe's accumulate truncates: a long chain of MFMAs into one accumulator drifts
  // (same-sign terms, as a bias gradient has them: 1.5e-5 of the result over the ~3 000
  // instructions of a wave's share at C4).  So the chain is cut every WG_FLUSH groups: the
  // accumulators go to running totals by vector adds (round to nearest) and start again from zero.
  // The totals live in LDS (16 words per lane): in registers they would cost the sixth wave per SIMD.
  __shared__ float tot[16 * 64];
  f32x4v acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    acc[b] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) tot[(4 * b + r) * 64 + lane] = 0.f;
  }
  const int e_own = n & 7;
  // the neighbours' mask words sit where the padding columns of the row are: columns 7 (low word:
  // gradient columns 4 q + cc, cc = 0, 1) and 15 (cc = 2, 3); bit 16 (cc & 1) + q
  const int msrc = 4 * (16 * kg + ((n & 2) ? 15 : 7));
  const int bit0 = 16 * (n & 1) + (n >> 2);               // + 4 b for column 16 b + n
  const bool special = n == 7 || n == 15;
  const float spec = n == 7 ? WG_TOP : 0.f;               // the bias column is the constant 1 (scaled)
  const uint32_t u_lane = 4u * (uint32_t)n;

  // this wave's groups, in order: blocks blockIdx.x, blockIdx.x + gridDim.x, ... of WG_BLOCK groups
  const int my_blocks = (int)blockIdx.x < nblocks ? (nblocks - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nit = my_blocks * WG_BLOCK;
  auto group_of = [&](int it) {
    return ((int)blockIdx.x + (it / WG_BLOCK) * (int)gridDim.x) * WG_BLOCK + (it & (WG_BLOCK - 1));
  };

#define WG_LOAD_DESC(DST, IT)                                              \
  {                                                                        \
    DST = make_int2(0, 0);                                                 \
    if ((IT) < nit) {                                                      \
      const int s_ = 4 * group_of(IT) + kg;                                \
      if (s_ < S) DST = desc[s_];                                          \
    }                                                                      \
  }
#define WG_LOAD_COL(DST, D)                                                \
  {                                                                        \
    DST = -1;                                                              \
    if (e_own < ((D).y & 15)) DST = col[(D).x + e_own];                    \
  }
#define WG_LOAD_ROWS(V, GQ, IDOWN, D)                                      \
  {                                                                        \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                        \
      const int id_ = __shfl(IDOWN, 16 * kg + e, 64);                      \
      V[e] = 0u;                                                           \
      if (id_ >= 0) V[e] = *(const uint32_t *)((const char *)U + ((uint32_t)id_ * 64u + u_lane)); \
    }                                                                      \
    const float *g_ = ga + (int64_t)((D).y >> 5) * 64 + n;                 \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) GQ[b] = ((D).y & 15) ? g_[16 * b] : 0.f; \
  }

  int2 d0, d1, d2, d3;
  int id1, id2;
  uint32_t v[8];
  float gq[4];
  {
    int id0;
    WG_LOAD_DESC(d0, 0)
    WG_LOAD_DESC(d1, 1)
    WG_LOAD_DESC(d2, 2)
    WG_LOAD_COL(id0, d0)
    WG_LOAD_COL(id1, d1)
    WG_LOAD_ROWS(v, gq, id0, d0)
  }
  for (int it = 0; it < nit; ++it) {
    uint32_t nv[8];
    float ngq[4];
    WG_LOAD_ROWS(nv, ngq, id1, d1)
    WG_LOAD_COL(id2, d2)
    WG_LOAD_DESC(d3, it + 3)
    // ---- A operand: uu[j_e][n] scaled into fp16 range and cut in two (hi = rtz16(v), lo = rtz16(v - hi))
    uint32_t wv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wv[e] = (uint32_t)__builtin_amdgcn_ds_bpermute(msrc, (int)v[e]);
    f16x8 a1, a2;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
      // (a neighbour that is not there has an empty mask: its column of B is zero whatever A holds)
      const float v0 = special ? spec : __uint_as_float(v[e]) * su;
      const float v1 = special ? spec : __uint_as_float(v[e + 1]) * su;
      const f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0, v1));
      const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0 - (float)hi[0], v1 - (float)hi[1]));
      a1[e] = hi[0]; a1[e + 1] = hi[1];
      a2[e] = lo[0]; a2[e + 1] = lo[1];
    }
    // ---- B operand per block of 16 gradient columns: the slot's gradient value where the
    // neighbour's bit is set
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float gs = gq[b] * sg;
      const f16x2 gh2 = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(gs, gs));
      const float gr = gs - (float)gh2[0];
      const uint32_t ghi = __builtin_bit_cast(uint32_t, gh2);
      const uint32_t glo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(gr, gr));
      u32x4v b1, b2;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_sbfe((int)wv[2 * q], bit0 + 4 * b, 1);
        const uint32_t t1 = (uint32_t)__builtin_amdgcn_sbfe((int)wv[2 * q + 1], bit0 + 4 * b, 1);
#ifdef WG_EXP_NOMASK
        const uint32_t m = t0 | t1 | 1u;
        b1[q] = m | ghi;
        b2[q] = m | glo;
#else
        const uint32_t m = (t0 & 0xFFFFu) | (t1 & 0xFFFF0000u);
        b1[q] = m & ghi;
        b2[q] = m & glo;
#endif
      }
      const f16x8 bh = __builtin_bit_cast(f16x8, b1), bl = __builtin_bit_cast(f16x8, b2);
      acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bh, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh, acc[b], 0, 0, 0);
    }
    if ((it & (WG_FLUSH - 1)) == WG_FLUSH - 1) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[(4 * b + r) * 64 + lane] += acc[b][r];
        acc[b] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      }
    }
    d0 = d1; d1 = d2; d2 = d3;
    id1 = id2;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = nv[e];
#pragma unroll
    for (int b = 0; b < 4; ++b) gq[b] = ngq[b];
  }
#undef WG_LOAD_DESC
#undef WG_LOAD_COL
#undef WG_LOAD_ROWS
  // ---- phase 2, the root terms: group gr = the listed rows 32 gr .. 32 gr + 31 as 32 "neighbours"
  // (j = the row itself), each against its OWN gradient row gx[j]; wave w takes gr = w, w + W, ...
  {
    int Rn = *n_rows;
    if (Rn > rows_cap) Rn = rows_cap;
    const int RG = (Rn + 31) >> 5;
    auto own_row = [&](int gr) {
      const int t_ = 32 * gr + 8 * kg + e_own;
      return gr < RG && t_ < Rn ? rows[t_] : -1;
    };
    int rid = own_row((int)blockIdx.x), trip = 0;
    for (int gr = blockIdx.x; gr < RG; gr += gridDim.x, ++trip) {
      const int rid_next = own_row(gr + (int)gridDim.x);
      int id[8];
      uint32_t uv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        id[e] = __shfl(rid, 16 * kg + e, 64);
        uv[e] = 0u;
        if (id[e] >= 0) uv[e] = *(const uint32_t *)((const char *)U + ((uint32_t)id[e] * 64u + u_lane));
      }
      uint32_t wv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = (uint32_t)__builtin_amdgcn_ds_bpermute(msrc, (int)uv[e]);
      f16x8 a1, a2;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const float v0 = special ? spec : __uint_as_float(uv[e]) * su;
        const float v1 = special ? spec : __uint_as_float(uv[e + 1]) * su;
        const f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0, v1));
        const f16x2 lo = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(v0 - (float)hi[0], v1 - (float)hi[1]));
        a1[e] = hi[0]; a1[e + 1] = hi[1];
        a2[e] = lo[0]; a2[e + 1] = lo[1];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // (fetched block by block: with the next block's values in flight as well the kernel spills)
        float gc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gc[e] = id[e] >= 0 ? gx[(int64_t)id[e] * 64 + 16 * b + n] : 0.f;
        u32x4v b1, b2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float s0 = gc[2 * q] * sg, s1 = gc[2 * q + 1] * sg;
          const f16x2 hi = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(s0, s1));
          const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(s0 - (float)hi[0], s1 - (float)hi[1]));
          const uint32_t t0 = (uint32_t)__builtin_amdgcn_sbfe((int)wv[2 * q], bit0 + 4 * b, 1);
          const uint32_t t1 = (uint32_t)__builtin_amdgcn_sbfe((int)wv[2 * q + 1], bit0 + 4 * b, 1);
          const uint32_t m = (t0 & 0xFFFFu) | (t1 & 0xFFFF0000u);
          b1[q] = m & __builtin_bit_cast(uint32_t, hi);
          b2[q] = m & lo;
        }
        const f16x8 bh = __builtin_bit_cast(f16x8, b1), bl = __builtin_bit_cast(f16x8, b2);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, bh, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh, acc[b], 0, 0, 0);
      }
      if ((trip & (WG_FLUSH - 1)) == WG_FLUSH - 1) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[(4 * b + r) * 64 + lane] += acc[b][r];
          acc[b] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
      }
      rid = rid_next;
    }
  }
  float *dst = partial + (int64_t)blockIdx.x * 1024;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * kg + r) * 64 + 16 * b + n] = tot[(4 * b + r) * 64 + lane] + acc[b][r];
}

// partial[w][1024] -> red[s][1024], s = w / (WG_WAVES / WG_SLICES), waves in ascending order
__global__ void __launch_bounds__(256) wg_reduce1_k(const float *__restrict__ partial, float *__restrict__ red) {
  const int el = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  constexpr int PER = WG_WAVES / WG_SLICES;
  const float *src = partial + (int64_t)s * PER * 1024 + el;
  float a = 0.f;
#pragma unroll 8
  for (int w = 0; w < PER; ++w) a += src[(int64_t)w * 1024];
  red[s * 1024 + el] = a;
}

// red[s][k][c] summed over s in order, unscaled, into grad_w_l [64][8], grad_b [64], grad_w_r [64][8]
__global__ void __launch_bounds__(256) wg_reduce2_k(const float *__restrict__ red, const float *__restrict__ params,
                                                    float *__restrict__ gwl, float *__restrict__ gb,
                                                    float *__restrict__ gwr) {
  const int el = blockIdx.x * 256 + threadIdx.x;
  float a = 0.f;
  for (int s = 0; s < WG_SLICES; ++s) a += red[s * 1024 + el];
  const int k = el >> 6, c = el & 63;
  if (k < 7) gwl[c * 8 + k] = a * params[2];
  else if (k == 7) { gb[c] = a * params[3]; gwl[c * 8 + 7] = 0.f; }
  else if (k < 15) gwr[c * 8 + (k - 8)] = a * params[2];
  else gwr[c * 8 + 7] = 0.f;
}

struct WgradWs {
  int *soff;         // rows_cap + 1 (counts, scanned in place) ...
  int *bsum;         // ... and the scan's block sums
  int2 *desc;
  float *params;
  float *partial;    // [WG_WAVES][16][64]
  float *red;        // [WG_SLICES][16][64]
  size_t bytes;
};
// a listed row has ceil(deg / 8) <= 1 + deg / 8 slots
static inline int64_t wg_max_slots(int64_t cap, int64_t E) { return cap + E / 8 + 4; }
static inline WgradWs wg_carve(void *buf, int64_t cap, int64_t E) {
  GsCarver c(buf);
  WgradWs w;
  w.soff = c.take<int>((size_t)cap + 1);
  w.bsum = c.take<int>((size_t)((cap + 1 + GS_SCAN_CH - 1) / GS_SCAN_CH + 1));
  w.desc = c.take<int2>((size_t)wg_max_slots(cap, E));
  w.params = c.take<float>(8);
  w.partial = c.take<float>((size_t)WG_WAVES * 1024);
  w.red = c.take<float>((size_t)WG_SLICES * 1024);
  w.bytes = c.off;
  return w;
}

}  // namespace

extern "C" int gsage_hidden1_wgrad_supported(int64_t Fo, int epilogue) {
  return Fo == 64 && epilogue == GSAGE_EPI_RELU_DROPOUT ? 1 : 0;
}

extern "C" size_t gsage_hidden1_wgrad_workspace_bytes(int64_t rows_cap, int64_t E) {
  if (rows_cap < 0 || E < 0) return 0;
  return wg_carve(nullptr, rows_cap, E).bytes;
}

extern "C" int gsage_hidden1_wgrad(const int32_t *rowptr, const int32_t *col, int64_t N, int64_t E,
                                   const float *u_rows, const float *grad_agg, const float *grad_x,
                                   const uint32_t *g_amax, const int32_t *rows, const int32_t *n_rows,
                                   int64_t rows_cap, int64_t Fo, int epilogue, float p, float *grad_w_l,
                                   float *grad_b, float *grad_w_r, void *ws, size_t ws_bytes,
                                   gsage_stream_t stream) {
  if (N < 0 || E < 0 || rows_cap <= 0 || !rowptr || !u_rows || !grad_agg || !grad_x || !g_amax || !rows ||
      !n_rows || !grad_w_l || !grad_b || !grad_w_r || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (E > 0 && !col) return GSAGE_ERR_INVALID_ARG;
  if (!gsage_hidden1_wgrad_supported(Fo, epilogue)) return GSAGE_ERR_UNSUPPORTED;
  if (!(p >= 0.f && p < 1.f)) return GSAGE_ERR_INVALID_ARG;
  // row numbers ride in 26 bits of a descriptor, rows of the table are addressed by 32-bit offsets,
  // slots are counted in an int
  if (N >= ((int64_t)1 << 26) || E >= ((int64_t)1 << 31) || rows_cap > N ||
      wg_max_slots(rows_cap, E) >= ((int64_t)1 << 31))
    return GSAGE_ERR_UNSUPPORTED;
  if ((((uintptr_t)u_rows | (uintptr_t)grad_agg | (uintptr_t)grad_x | (uintptr_t)ws) & 15) != 0)
    return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < gsage_hidden1_wgrad_workspace_bytes(rows_cap, E)) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  const WgradWs w = wg_carve(ws, rows_cap, E);
  const uint32_t *u_amax = (const uint32_t *)(u_rows + (size_t)N * 16);     // first word of row N
  hipLaunchKernelGGL(wg_params_k, dim3(1), dim3(1), 0, st, u_amax, g_amax, 1.0f / (1.0f - p), w.params);
  GSAGE_CHECK_LAUNCH();
  const unsigned cb = (unsigned)((rows_cap + 1 + 255) / 256);
  hipLaunchKernelGGL(wg_count_k, dim3(cb), dim3(256), 0, st, rows, n_rows, rows_cap, rowptr, w.soff);
  GSAGE_CHECK_LAUNCH();
  if (int rc = gs_exclusive_scan(w.soff, rows_cap + 1, w.soff, w.bsum, st)) return rc;
  hipLaunchKernelGGL(wg_fill_k, dim3(cb), dim3(256), 0, st, rows, n_rows, rows_cap, rowptr,
                     (const int *)w.soff, w.desc);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(hidden1_wgrad_k, dim3(WG_WAVES), dim3(64), 0, st, (const int2 *)w.desc,
                     (const int *)(w.soff + rows_cap), col, u_rows, grad_agg, grad_x, (const float *)w.params,
                     w.partial, rows, n_rows, (int)rows_cap);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(wg_reduce1_k, dim3(4, WG_SLICES), dim3(256), 0, st, (const float *)w.partial, w.red);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(wg_reduce2_k, dim3(4), dim3(256), 0, st, (const float *)w.red, (const float *)w.params,
                     grad_w_l, grad_b, grad_w_r);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}
