// K4 / K5: scaled skip-gram negative-sampling loss for gfx950 (HBM-bound row gathers).
//
// Forward: 16 lanes cover one embedding row with 16-byte loads (D <= 256, D % 4 == 0),
// so a wavefront works on 4 pairs at once; the target row stays in registers while the
// context row and the K negative rows stream past it (6 rows in flight per lane).  The
// six reduced quantities are accumulated per lane, combined per workgroup, written as
// per-workgroup partials and summed in double in a fixed order by a second tiny kernel:
// the loss value is bitwise reproducible.  d(loss)/d(score) is stored per pair so the
// backward does not recompute any dot product.
//
// Backward: one wavefront per pair with lane = embedding column, so every atomic
// wave-instruction adds 256 contiguous bytes (the shape that runs at the full
// float-atomic rate on MI355X).
#include "gsage_common.h"
#include "gsage_sort.h"

#include <mutex>

namespace {

constexpr int SG_MAX_BLOCKS = 2048;
constexpr int SG_NSTAT = 6;

__device__ __forceinline__ float softplus_neg(float x) {  // -logsigmoid(x)
  return fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

struct SgArgs {
  const float *emb;
  int64_t N, D;
  const int64_t *tg, *ctx, *neg;
  int64_t B, K;
  float scale, invB, invBK;
  float *coef;
  float *partials;  // [gridDim.x][8]
};

// slot 6 of a workgroup's partials carries its "saw an out-of-range id" flag: no shared flag
// word, so nothing has to be zeroed before the launch (a hipMemsetAsync node in front of the
// kernel was seen to leave the word stale about once in 2 500 replays of a captured HIP graph)
__device__ __forceinline__ void sg_block_reduce(float (&st)[SG_NSTAT], bool bad, float *partials) {
  __shared__ float red[4][SG_NSTAT + 1];
  const int lane = gs_lane(), w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < SG_NSTAT; ++i) {
    float v = gs_wave_sum(st[i]);
    if (lane == 0) red[w][i] = v;
  }
  const bool wave_bad = __ballot(bad) != 0ull;
  if (lane == 0) red[w][SG_NSTAT] = wave_bad ? 1.f : 0.f;
  __syncthreads();
  if (threadIdx.x <= SG_NSTAT)
    partials[blockIdx.x * 8 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] +
                                             red[2][threadIdx.x] + red[3][threadIdx.x];
}

__device__ __forceinline__ void sg_terms(const SgArgs &a, int64_t b, int r, float raw,
                                         float (&st)[SG_NSTAT]) {
  float sc = raw * a.scale;
  float sg = sigmoidf(sc);
  if (r == 0) {
    st[0] += softplus_neg(sc);
    st[2] += raw;
    st[4] += sg;
    a.coef[b * (1 + a.K)] = -a.scale * sigmoidf(-sc) * a.invB;
  } else {
    st[1] += softplus_neg(-sc);
    st[3] += raw;
    st[5] += sg;
    a.coef[b * (1 + a.K) + r] = a.scale * sg * a.invBK;
  }
}

template <int DCH>
__global__ void __launch_bounds__(256) skipgram_fwd_vec_k(SgArgs a) {
  const int lane = gs_lane(), sub = lane & 15, grp = lane >> 4;
  const int64_t D = a.D, K = a.K, R = 1 + a.K;
  float st[SG_NSTAT] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool bad = false;
  const int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t base = first; base < a.B; base += stride) {
    const int64_t b = base + grp;
    const bool active = b < a.B;
    int64_t t = active ? a.tg[b] : 0;
    const bool tbad = (uint64_t)t >= (uint64_t)a.N;
    if (tbad) { bad |= active; t = 0; }
    float4 tr[DCH];
#pragma unroll
    for (int c = 0; c < DCH; ++c) {
      int d = c * 64 + sub * 4;
      tr[c] = d < D ? *(const float4 *)(a.emb + t * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t r0 = 0; r0 < R; r0 += 6) {
      float dotv[6];
      bool ok[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        int64_t r = r0 + u;
        bool valid = active && !tbad && r < R;
        int64_t idx = 0;
        if (valid) idx = r == 0 ? a.ctx[b] : a.neg[b * K + r - 1];
        if ((uint64_t)idx >= (uint64_t)a.N) { bad |= valid; valid = false; idx = 0; }
        ok[u] = valid;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DCH; ++c) {
          int d = c * 64 + sub * 4;
          if (d < D) {
            float4 e = *(const float4 *)(a.emb + idx * D + d);
            s += tr[c].x * e.x + tr[c].y * e.y + tr[c].z * e.z + tr[c].w * e.w;
          }
        }
        dotv[u] = s;
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        float s = dotv[u];
        dotv[u] = row16_sum(s);      // (DPP row rotations: bitwise the xor butterfly's sum)
      }
      // after the xor-reduction every lane of the group holds all 6 dot products: lane u of the
      // group finishes term u (sigmoid, softplus, coefficient store) instead of lane 0 doing
      // six in a row
      {
        float myv = dotv[0];
        bool myok = ok[0];
#pragma unroll
        for (int u = 1; u < 6; ++u) {
          myv = sub == u ? dotv[u] : myv;
          myok = sub == u ? ok[u] : myok;
        }
        const int64_t r = r0 + sub;
        if (sub < 6) {
          if (myok) sg_terms(a, b, (int)r, myv, st);
          else if (active && r < R) a.coef[b * R + r] = 0.f;
        }
      }
    }
  }
  sg_block_reduce(st, bad, a.partials);
}

// any D: one wavefront per pair, lanes stride the columns
__global__ void __launch_bounds__(256) skipgram_fwd_generic_k(SgArgs a) {
  const int lane = gs_lane();
  const int64_t D = a.D, K = a.K, R = 1 + a.K;
  float st[SG_NSTAT] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bool bad = false;
  const int64_t first = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t b = first; b < a.B; b += stride) {
    int64_t t = a.tg[b];
    bool tbad = (uint64_t)t >= (uint64_t)a.N;
    if (tbad) { bad = true; t = 0; }
    for (int64_t r = 0; r < R; ++r) {
      int64_t idx = r == 0 ? a.ctx[b] : a.neg[b * K + r - 1];
      bool valid = !tbad;
      if ((uint64_t)idx >= (uint64_t)a.N) { bad = true; valid = false; idx = 0; }
      float s = 0.f;
      for (int64_t d = lane; d < D; d += 64) s = fmaf(a.emb[t * D + d], a.emb[idx * D + d], s);
      s = gs_wave_sum(s);
      if (lane == 0) {
        if (valid) sg_terms(a, b, (int)r, s, st);
        else a.coef[b * R + r] = 0.f;
      }
    }
  }
  sg_block_reduce(st, bad, a.partials);
}

// ordered double-precision sum of the per-workgroup partials: wave i sums statistic i (lanes
// stride the workgroups, then a fixed xor tree), lane 0 of wave 0 finishes -- the order of every
// addition is fixed, so the loss is bitwise reproducible
__global__ void __launch_bounds__(512)
skipgram_finish_k(const float *__restrict__ partials, int nblocks, float invB, float invBK,
                  float *__restrict__ stats) {
  __shared__ double tot[SG_NSTAT + 1];
  const int lane = gs_lane(), i = threadIdx.x >> 6;
  if (i <= SG_NSTAT) {
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int b = lane;
    for (; b + 192 < nblocks; b += 256) {
      v0 += (double)partials[b * 8 + i];
      v1 += (double)partials[(b + 64) * 8 + i];
      v2 += (double)partials[(b + 128) * 8 + i];
      v3 += (double)partials[(b + 192) * 8 + i];
    }
    for (; b < nblocks; b += 64) v0 += (double)partials[b * 8 + i];
    double v = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) tot[i] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double *s = tot;
    float pos_loss = (float)(s[0] * (double)invB);
    float neg_loss = (float)(s[1] * (double)invBK);
    stats[0] = pos_loss + neg_loss;
    stats[1] = pos_loss;
    stats[2] = neg_loss;
    stats[3] = (float)(s[2] * (double)invB);
    stats[4] = (float)(s[3] * (double)invBK);
    stats[5] = (float)(s[4] * (double)invB);
    stats[6] = (float)(s[5] * (double)invBK);
    stats[7] = s[SG_NSTAT] > 0.0 ? (float)GSAGE_FLAG_INDEX_OUT_OF_RANGE : 0.f;
  }
}

__global__ void __launch_bounds__(256)
skipgram_bwd_k(const float *__restrict__ emb, int64_t N, int64_t D, const int64_t *__restrict__ tg,
               const int64_t *__restrict__ ctx, const int64_t *__restrict__ neg, int64_t B,
               int64_t K, const float *__restrict__ coef, const float *__restrict__ grad_loss,
               float *__restrict__ grad_emb) {
  const int lane = gs_lane();
  const int64_t R = 1 + K;
  const float g = *grad_loss;
  const int64_t first = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t b = first; b < B; b += stride) {
    const int64_t t = tg[b];
    if ((uint64_t)t >= (uint64_t)N) continue;
    for (int64_t d = lane; d < D; d += 64) {
      const float tv = emb[t * D + d];
      float gt = 0.f;
      for (int64_t r = 0; r < R; ++r) {
        int64_t idx = r == 0 ? ctx[b] : neg[b * K + r - 1];
        if ((uint64_t)idx >= (uint64_t)N) continue;
        float cf = coef[b * R + r] * g;
        gt = fmaf(cf, emb[idx * D + d], gt);
        atomicAdd(grad_emb + idx * D + d, cf * tv);
      }
      atomicAdd(grad_emb + t * D + d, gt);
    }
  }
}


// ------------------------------------------------------- deterministic backward (no atomics)
// Every pair contributes 2(1+K) terms  grad_emb[dst] += w * emb[src].  The terms are listed,
// stably sorted by dst with the K1 radix sort (term id as payload, so equal dst keep term
// order), turned into a per-batch CSR and summed by the segment-gather kernel with per-term
// weights: grad_emb is written row by row in a fixed order => bitwise reproducible, and no
// zero-fill pass is needed.
//   term s of pair b, R = 1+K:  s <  R: dst = t_b,            src = role(s)   (c_b or n_b,s-1)
//                               s >= R: dst = role(s-R),      src = t_b        weight coef[b][s mod R]
__device__ __forceinline__ void sg_term(const int64_t *__restrict__ tg, const int64_t *__restrict__ ctx,
                                        const int64_t *__restrict__ neg, int64_t K, int64_t b, int s,
                                        int64_t &dst, int64_t &src, int &r) {
  const int R = 1 + (int)K;
  r = s < R ? s : s - R;
  int64_t t = tg[b];
  int64_t o = r == 0 ? ctx[b] : neg[b * K + r - 1];
  dst = s < R ? t : o;
  src = s < R ? o : t;
}

__global__ void __launch_bounds__(256)
sg_terms_k(const int64_t *__restrict__ tg, const int64_t *__restrict__ ctx,
           const int64_t *__restrict__ neg, int64_t B, int64_t K, int64_t N,
           const float *__restrict__ coef, const float *__restrict__ grad_loss,
           int *__restrict__ key, int *__restrict__ tsrc, float *__restrict__ tw) {
  const int64_t M = B * 2 * (1 + K);
  const float g = *grad_loss;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M; e += stride) {
    int64_t b = e / (2 * (1 + K));
    int s = (int)(e % (2 * (1 + K)));
    int64_t dst, src;
    int r;
    sg_term(tg, ctx, neg, K, b, s, dst, src, r);
    bool bad = (uint64_t)dst >= (uint64_t)N || (uint64_t)src >= (uint64_t)N;
    int k = bad ? (int)N : (int)dst;          // sentinel row N: sorted last, never read
    key[e] = k;
    // source row and weight of the term travel through the (stable) sort as its two payloads: no
    // permutation to apply afterwards, and equal rows keep their terms in term order
    tsrc[e] = bad ? 0 : (int)src;
    tw[e] = bad ? 0.f : coef[b * (1 + K) + r] * g;
  }
}

struct DetLayout {
  size_t total, o_key, o_src, o_tw, o_kA, o_vA, o_wA, o_kB, o_vB, o_wB, o_counts, o_bsum, o_rowptr,
      o_col, o_w, o_long, o_nlong;
  int64_t M, long_cap;
};

DetLayout det_layout(int64_t N, int64_t B, int64_t K) {
  DetLayout L;
  L.M = B * 2 * (1 + K);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += gs_align_up(bytes ? bytes : 4, 256);
    return o;
  };
  size_t m4 = (size_t)L.M * 4;
  int64_t ntiles = (L.M + GS_RS_TILE - 1) / GS_RS_TILE;
  int64_t ncounts = 256 * (ntiles > 0 ? ntiles : 1);
  int64_t scan_n = ncounts > (N + 2) ? ncounts : (N + 2);
  L.long_cap = L.M / gsage_long_row_threshold() + 1;
  L.o_key = take(m4); L.o_src = take(m4); L.o_tw = take(m4);
  L.o_kA = take(m4); L.o_vA = take(m4); L.o_wA = take(m4);
  L.o_kB = take(m4); L.o_vB = take(m4); L.o_wB = take(m4);
  L.o_counts = take((size_t)ncounts * 4);
  L.o_bsum = take((size_t)((scan_n + GS_SCAN_CH - 1) / GS_SCAN_CH + 1) * 4);
  L.o_rowptr = take((size_t)(N + 2) * 4);
  L.o_col = take(m4);
  L.o_w = take(m4);
  L.o_long = take((size_t)L.long_cap * 4);
  L.o_nlong = take(16);
  L.total = off;
  return L;
}

// ------------------------------------------------------------ rows a pair batch touches
// d(loss)/d(emb) is non-zero only on the rows the batch names (targets, contexts, negatives).
// They are known before the backward starts: one bit per row, then the sorted list of the set
// bits, so that the layer behind the loss can walk the touched rows only.
__global__ void __launch_bounds__(256)
rows_mark_k(const int64_t *__restrict__ tg, const int64_t *__restrict__ ctx,
            const int64_t *__restrict__ neg, int64_t B, int64_t K, int64_t N,
            uint8_t *__restrict__ flags) {
  // one BYTE per row, set by plain stores: racing stores of the same value need no atomic, and
  // unlike a read-modify-write atomic a plain store leaves its line in the L2 (an atomicOr on a
  // bit mask measured 0.40 ms for 7.3 M ids at 10 M rows: every atomic drops the line it hit,
  // the next id on that line fetches it again -- 4.8 M line fetches for a 1.25 MB mask)
  const int64_t total = B * (2 + K);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t idx = e < B ? tg[e] : e < 2 * B ? ctx[e - B] : neg[e - 2 * B];
    if ((uint64_t)idx < (uint64_t)N) flags[idx] = 1;
  }
}

constexpr int RC_BLOCK = 256;   // mask words per workgroup (one per thread)

// 32 row flags -> one mask word (written here: the mask needs no clearing), popcounts per block
__global__ void __launch_bounds__(RC_BLOCK)
rows_count_k(const uint8_t *__restrict__ flags, int64_t nwords, uint32_t *__restrict__ mask,
             int *__restrict__ block_sums) {
  __shared__ int red[RC_BLOCK / 64];
  const int64_t i = (int64_t)blockIdx.x * RC_BLOCK + threadIdx.x;
  uint32_t word = 0u;
  if (i < nwords) {
    const uint4 lo = *(const uint4 *)(flags + i * 32), hi = *(const uint4 *)(flags + i * 32 + 16);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t v = w[q];          // four flag bytes, each 0 or 1
      word |= ((v & 1u) | ((v >> 7) & 2u) | ((v >> 14) & 4u) | ((v >> 21) & 8u)) << (4 * q);
    }
    mask[i] = word;
  }
  int c = __popc(word);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (gs_lane() == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one workgroup: exclusive scan of the block sums in place, total -> n_rows[0]
__global__ void __launch_bounds__(1024)
rows_scan_k(int *__restrict__ block_sums, int nblocks, int *__restrict__ n_rows) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + t;
    const int v = i < nblocks ? block_sums[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int before = carry_s;
    for (int k = 0; k < w; ++k) before += wsum[k];
    if (i < nblocks) block_sums[i] = before + inc - v;
    __syncthreads();
    if (t == 1023) carry_s = before + inc;
    __syncthreads();
  }
  if (t == 0) n_rows[0] = carry_s;
}

__global__ void __launch_bounds__(RC_BLOCK)
rows_emit_k(const uint32_t *__restrict__ mask, int64_t nwords, const int *__restrict__ block_off,
            int *__restrict__ rows, int64_t cap, int *__restrict__ word_prefix) {
  __shared__ int wsum[RC_BLOCK / 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t i = (int64_t)blockIdx.x * RC_BLOCK + t;
  uint32_t m = i < nwords ? mask[i] : 0u;
  const int c = __popc(m);
  int inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int64_t pos = block_off[blockIdx.x] + inc - c;
  for (int k = 0; k < w; ++k) pos += wsum[k];
  // place in the list of the word's first set bit: the rank of row r is
  // word_prefix[r >> 5] + popcount(mask[r >> 5] & ((1 << (r & 31)) - 1))
  if (word_prefix && i < nwords) word_prefix[i] = (int)pos;
  const int row0 = (int)(i << 5);
  while (m) {
    const int b = __ffs(m) - 1;
    m &= m - 1;
    if (pos < cap) rows[pos] = row0 + b;
    ++pos;
  }
}

// zero `n_rows` rows of D floats each, named by a row list (4 rows of 64 floats per wave store)
__global__ void __launch_bounds__(256)
rows_zero_k(const int *__restrict__ rows, const int *__restrict__ n_rows, int64_t cap, int64_t D,
            float *__restrict__ buf) {
  int64_t n = *n_rows;
  if (n > cap) n = cap;
  const int64_t d4 = D / 4;                       // D % 4 == 0 checked by the caller
  const int64_t total = n * d4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t r = rows[e / d4];
    *(float4 *)(buf + r * D + (e % d4) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// ------------------------------------- deterministic backward over the touched rows only
// The sorted form above scales with N (its row boundaries cover all N rows, its radix keys have
// log2 N bits, its gather writes N rows): at 10 M rows and 2^20 pairs it loses to the atomics.
// With the touched-row list in hand the key of a term is the PLACE of its destination row in that
// list (word prefix + popcount below the row's bit: two L2-resident reads), the per-batch CSR has
// one row per listed row, and the gather writes only those rows of grad_emb -- each exactly once,
// with its whole sum, in term order: bitwise reproducible, nothing to zero beforehand.
__global__ void __launch_bounds__(256)
sg_terms_rows_k(const int64_t *__restrict__ tg, const int64_t *__restrict__ ctx,
                const int64_t *__restrict__ neg, int64_t B, int64_t K, int64_t N,
                const float *__restrict__ coef, const float *__restrict__ grad_loss,
                const uint32_t *__restrict__ mask, const int *__restrict__ word_prefix, int sentinel,
                int *__restrict__ key, int *__restrict__ tsrc, float *__restrict__ tw) {
  const int64_t M = B * 2 * (1 + K);
  const float g = *grad_loss;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M; e += stride) {
    int64_t b = e / (2 * (1 + K));
    int s = (int)(e % (2 * (1 + K)));
    int64_t dst, src;
    int r;
    sg_term(tg, ctx, neg, K, b, s, dst, src, r);
    const bool bad = (uint64_t)dst >= (uint64_t)N || (uint64_t)src >= (uint64_t)N;
    int k = sentinel;                          // sorted last, belongs to no listed row
    if (!bad) {
      const uint32_t m = mask[dst >> 5];
      const int bit = (int)(dst & 31);
      k = word_prefix[dst >> 5] + __popc(m & ((1u << bit) - 1u));
      if (!((m >> bit) & 1u) || k >= sentinel) k = sentinel;   // (a row the mask does not name)
    }
    // (row, source, weight) travel through the sort together: it is stable, so the terms of a
    // row stay in term order without a term number
    key[e] = k;
    tsrc[e] = bad ? 0 : (int)src;
    tw[e] = bad ? 0.f : coef[b * (1 + K) + r] * g;
  }
}

// out[rows[r], :] = sum over the terms p of listed row r of w[p] * emb[col[p], :], r < *n_rows.
// A wave takes G = 64 / LPR consecutive listed rows as ONE flat run of terms (a listed row has 2.4
// terms on average at the headline batch): 64 terms per trip whatever the row boundaries, each
// lane group fetching every G-th one with NB loads in flight and adding it to the accumulator of
// the row it belongs to; one full-wave store at the end.  Term order within a row is the sort's
// (stable: term number), the lane group and trip a term lands in follow from its position alone.
template <int LPR>
__global__ void __launch_bounds__(256)
sg_rows_gather_k(const int *__restrict__ rowptr, const int *__restrict__ col,
                 const float *__restrict__ w, const float *__restrict__ emb,
                 const int *__restrict__ rows, const int *__restrict__ n_rows, int64_t cap,
                 int64_t D, float *__restrict__ out) {
  constexpr int G = 64 / LPR;
#ifndef SG_ROWS_NB
#define SG_ROWS_NB 4
#endif
  constexpr int NB = SG_ROWS_NB;
  const int lane = gs_lane();
  const int slot = lane / LPR, sub = lane % LPR;
  int64_t n = *n_rows;
  if (n > cap) n = cap;
  const int64_t rg = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G;
  if (rg >= n) return;
  int b[G + 1];
#pragma unroll
  for (int i = 0; i <= G; ++i) b[i] = rowptr[rg + i < n ? rg + i : n];
  const int64_t my_row = rg + slot < n ? rows[rg + slot] : -1;
  float4 acc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int pend = b[G];
  for (int pc = b[0]; pc < pend; pc += 64) {
    const int pl = pc + lane;
    const bool have = pl < pend;
    const int cv = col[have ? pl : pend - 1];
    const float wv = have ? w[pl] : 0.f;
    int rv = 0;
#pragma unroll
    for (int i = 1; i < G; ++i) rv += pl >= b[i] ? 1 : 0;
    int cnt = pend - pc;
    if (cnt > 64) cnt = 64;
    const int steps = (cnt + G - 1) / G;
    for (int s = 0; s < steps; s += NB) {
      float4 r[NB];
      float wt[NB];
      int ri[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = ((s + k) * G + slot) & 63;
        const bool valid = (s + k) * G + slot < cnt;
        const int c = __shfl(cv, j, 64);
        wt[k] = __shfl(wv, j, 64);
        const int rj = __shfl(rv, j, 64);      // (every lane takes part: lane j may itself be past the end)
        ri[k] = valid ? rj : -1;
        // whole steps past the end (the same for every lane) fetch nothing; within the last step the
        // slots past the end re-read the last term's row (cached) and are answered by no accumulator
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s + k < steps) r[k] = *(const float4 *)(emb + (int64_t)c * D + sub * 4);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (s + k >= steps) break;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const bool hit = ri[k] == i;          // select the product, not the weight: 0 * inf is NaN
          acc[i].x += hit ? r[k].x * wt[k] : 0.f; acc[i].y += hit ? r[k].y * wt[k] : 0.f;
          acc[i].z += hit ? r[k].z * wt[k] : 0.f; acc[i].w += hit ? r[k].w * wt[k] : 0.f;
        }
      }
    }
  }
  float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < G; ++i) {
    float4 t = acc[i];
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64);
      t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
    }
    if (slot == i) keep = t;
  }
  if (my_row >= 0) *(float4 *)(out + my_row * D + sub * 4) = keep;
}

struct RowsLayout {
  size_t total, o_key, o_src, o_w, o_kA, o_vA, o_wA, o_kB, o_vB, o_wB, o_counts, o_bsum, o_rowptr,
      o_col, o_cw;
  int64_t M;
};

RowsLayout rows_layout(int64_t rows_cap, int64_t B, int64_t K) {
  RowsLayout L;
  L.M = B * 2 * (1 + K);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += gs_align_up(bytes ? bytes : 4, 256);
    return o;
  };
  const size_t m4 = (size_t)L.M * 4;
  const int64_t ntiles = (L.M + GS_RS_TILE - 1) / GS_RS_TILE;
  const int64_t ncounts = 256 * (ntiles > 0 ? ntiles : 1);
  const int64_t scan_n = ncounts > (rows_cap + 2) ? ncounts : (rows_cap + 2);
  L.o_key = take(m4); L.o_src = take(m4); L.o_w = take(m4);
  L.o_kA = take(m4); L.o_vA = take(m4); L.o_wA = take(m4);
  L.o_kB = take(m4); L.o_vB = take(m4); L.o_wB = take(m4);
  L.o_counts = take((size_t)ncounts * 4);
  L.o_bsum = take((size_t)((scan_n + GS_SCAN_CH - 1) / GS_SCAN_CH + 1) * 4);
  L.o_rowptr = take((size_t)(rows_cap + 2) * 4);
  L.o_col = take(m4);
  L.o_cw = take(m4);
  L.total = off;
  return L;
}

// Grid of the (grid-stride) forward kernels: never more workgroups than the chip holds at once.
// 256 CUs x 6 workgroups (the vector kernel runs 7 waves per SIMD): with 2048 workgroups the last
// 256 started only when the first 1792 had finished their 32 trips, i.e. the kernel ran two
// rounds where 1.14 would do: 0.56 -> 0.43 ms at C4 (5 per CU: 0.48, 7 per CU: 0.62).
#ifndef SG_GRID_CAP
#define SG_GRID_CAP (256 * 6)
#endif
int sg_blocks(int64_t B, int pairs_per_block) {
  int64_t nb = (B + pairs_per_block - 1) / pairs_per_block;
  if (nb < 1) nb = 1;
  if (nb > SG_GRID_CAP) nb = SG_GRID_CAP;
  return (int)nb;
}
static_assert(SG_GRID_CAP <= SG_MAX_BLOCKS, "partials are sized for SG_MAX_BLOCKS workgroups");

}  // namespace

extern "C" size_t gsage_skipgram_workspace_bytes(int64_t B) {
  (void)B;
  return (size_t)SG_MAX_BLOCKS * 8 * sizeof(float);
}

extern "C" int gsage_skipgram_fwd(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                                  const int64_t *contexts, const int64_t *negatives, int64_t B,
                                  int64_t K, float scale, double loss_norm_B, double loss_norm_BK,
                                  float *coef, float *stats, void *ws, size_t ws_bytes,
                                  gsage_stream_t stream) {
  if (N <= 0 || D <= 0 || B <= 0 || K < 0 || !emb || !targets || !contexts || !coef || !stats ||
      !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (K > 0 && !negatives) return GSAGE_ERR_INVALID_ARG;
  if (!(loss_norm_B > 0.0)) return GSAGE_ERR_INVALID_ARG;
  if (K > 0 && !(loss_norm_BK > 0.0)) return GSAGE_ERR_INVALID_ARG;
  if (ws_bytes < gsage_skipgram_workspace_bytes(B)) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  float *partials = (float *)ws;
  SgArgs a;
  a.emb = emb; a.N = N; a.D = D; a.tg = targets; a.ctx = contexts; a.neg = negatives;
  a.B = B; a.K = K; a.scale = scale;
  a.invB = (float)(1.0 / loss_norm_B);
  a.invBK = K > 0 ? (float)(1.0 / loss_norm_BK) : 0.f;
  a.coef = coef; a.partials = partials;
  int nb;
  const bool vec = (D % 4 == 0) && D <= 256 && (((uintptr_t)emb) & 15) == 0;
  if (vec) {
    nb = sg_blocks(B, 16);
    if (D <= 64) hipLaunchKernelGGL((skipgram_fwd_vec_k<1>), dim3(nb), dim3(256), 0, st, a);
    else if (D <= 128) hipLaunchKernelGGL((skipgram_fwd_vec_k<2>), dim3(nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((skipgram_fwd_vec_k<4>), dim3(nb), dim3(256), 0, st, a);
  } else {
    nb = sg_blocks(B, 4);
    hipLaunchKernelGGL(skipgram_fwd_generic_k, dim3(nb), dim3(256), 0, st, a);
  }
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(skipgram_finish_k, dim3(1), dim3(512), 0, st, partials, nb, a.invB, a.invBK,
                     stats);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

// ------------------------------------------------- K4 for callers that cannot allocate
// The per-workgroup partials of the loss forward live in a file-scope device array instead of a
// caller's workspace.  `scratch` below is a HOST-side stand-in for the symbol: its device
// address is looked up once per device with hipGetSymbolAddress (each GPU holds its own copy of
// the array at its own address) and only that pointer is ever handed to a kernel.
static __device__ __align__(256) float scratch[SG_MAX_BLOCKS * 8];   // 64 KiB per GPU

namespace {
constexpr int SG_MAX_DEVICES = 64;
std::mutex sg_scratch_mu;
float *sg_scratch_ptr[SG_MAX_DEVICES];   // per device; nullptr = not looked up yet

// device address of `scratch` on the CURRENT device (looked up on the first call there)
int sg_scratch_for_current_device(float **out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SG_MAX_DEVICES) return GSAGE_ERR_LAUNCH;
  std::lock_guard<std::mutex> lock(sg_scratch_mu);
  if (!sg_scratch_ptr[dev]) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) {
      (void)hipGetLastError();
      return GSAGE_ERR_LAUNCH;
    }
    sg_scratch_ptr[dev] = (float *)p;
  }
  *out = sg_scratch_ptr[dev];
  return GSAGE_OK;
}
}  // namespace

extern "C" int64_t gsage_skipgram_noalloc_max_pairs(void) { return GSAGE_SKIPGRAM_NOALLOC_MAX_PAIRS; }

extern "C" int gsage_skipgram_fwd_noalloc(const float *emb, int64_t N, int64_t D,
                                          const int64_t *targets, const int64_t *contexts,
                                          const int64_t *negatives, int64_t B, int64_t K,
                                          float scale, double loss_norm_B, double loss_norm_BK,
                                          float *coef, float *stats, gsage_stream_t stream) {
  if (B > GSAGE_SKIPGRAM_NOALLOC_MAX_PAIRS) return GSAGE_ERR_TOO_LARGE;
  float *ptr = nullptr;
  int rc = sg_scratch_for_current_device(&ptr);
  if (rc) return rc;
  // every workgroup overwrites its whole partial row and the finishing kernel reads only the
  // rows of this launch's grid, so nothing of the previous call's contents has to be cleared
  return gsage_skipgram_fwd(emb, N, D, targets, contexts, negatives, B, K, scale, loss_norm_B,
                            loss_norm_BK, coef, stats, ptr, sizeof(scratch), stream);
}

extern "C" size_t gsage_touched_rows_workspace_bytes(int64_t N) {
  if (N <= 0) return 0;
  const int64_t nwords = (N + 31) / 32;
  // per-block popcounts, then one flag byte per row (padded to whole mask words)
  return gs_align_up((size_t)((nwords + RC_BLOCK - 1) / RC_BLOCK) * sizeof(int), 256) +
         gs_align_up((size_t)nwords * 32, 256);
}

extern "C" int gsage_touched_rows(const int64_t *targets, const int64_t *contexts,
                                  const int64_t *negatives, int64_t B, int64_t K, int64_t N,
                                  uint32_t *mask, int32_t *rows, int64_t rows_cap, int32_t *n_rows,
                                  void *ws, size_t ws_bytes, gsage_stream_t stream) {
  return gsage_touched_rows_ex(targets, contexts, negatives, B, K, N, mask, rows, rows_cap, n_rows,
                               nullptr, ws, ws_bytes, stream);
}

extern "C" int gsage_touched_rows_ex(const int64_t *targets, const int64_t *contexts,
                                     const int64_t *negatives, int64_t B, int64_t K, int64_t N,
                                     uint32_t *mask, int32_t *rows, int64_t rows_cap,
                                     int32_t *n_rows, int32_t *word_prefix, void *ws,
                                     size_t ws_bytes, gsage_stream_t stream) {
  if (N <= 0 || B <= 0 || K < 0 || !targets || !contexts || !mask || !rows || !n_rows || !ws ||
      rows_cap <= 0)
    return GSAGE_ERR_INVALID_ARG;
  if (K > 0 && !negatives) return GSAGE_ERR_INVALID_ARG;
  if (N >= (1ll << 31) - 64) return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < gsage_touched_rows_workspace_bytes(N)) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  const int64_t nwords = (N + 31) / 32;
  const int nblocks = (int)((nwords + RC_BLOCK - 1) / RC_BLOCK);
  int *block_sums = (int *)ws;
  uint8_t *flags = (uint8_t *)ws + gs_align_up((size_t)nblocks * sizeof(int), 256);
  if (!gs_zero_async(flags, (size_t)nwords * 32, st)) return GSAGE_ERR_LAUNCH;
  hipLaunchKernelGGL(rows_mark_k, dim3(gs_grid(B * (2 + K), 256, 256 * 16)), dim3(256), 0, st,
                     targets, contexts, negatives, B, K, N, flags);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(rows_count_k, dim3(nblocks), dim3(RC_BLOCK), 0, st, flags, nwords, mask,
                     block_sums);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(rows_scan_k, dim3(1), dim3(1024), 0, st, block_sums, nblocks, n_rows);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(rows_emit_k, dim3(nblocks), dim3(RC_BLOCK), 0, st, mask, nwords, block_sums,
                     rows, rows_cap, word_prefix);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_zero_rows(const int32_t *rows, const int32_t *n_rows, int64_t rows_cap,
                               int64_t D, float *buf, gsage_stream_t stream) {
  if (!rows || !n_rows || !buf || rows_cap <= 0 || D <= 0 || D % 4 != 0 ||
      (((uintptr_t)buf) & 15) != 0)
    return GSAGE_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rows_zero_k, dim3(gs_grid(rows_cap * (D / 4), 256, 256 * 16)), dim3(256), 0,
                     gs_stream(stream), rows, n_rows, rows_cap, D, buf);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_skipgram_bwd(const float *emb, int64_t N, int64_t D, const int64_t *targets,
                                  const int64_t *contexts, const int64_t *negatives, int64_t B,
                                  int64_t K, const float *coef, const float *grad_loss,
                                  float *grad_emb, gsage_stream_t stream) {
  if (N <= 0 || D <= 0 || B <= 0 || K < 0 || !emb || !targets || !contexts || !coef ||
      !grad_loss || !grad_emb)
    return GSAGE_ERR_INVALID_ARG;
  if (K > 0 && !negatives) return GSAGE_ERR_INVALID_ARG;
  hipStream_t st = gs_stream(stream);
  int nb = gs_grid(B, 4, 256 * 16);
  hipLaunchKernelGGL(skipgram_bwd_k, dim3(nb), dim3(256), 0, st, emb, N, D, targets, contexts,
                     negatives, B, K, coef, grad_loss, grad_emb);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" size_t gsage_skipgram_bwd_det_workspace_bytes(int64_t N, int64_t B, int64_t K) {
  if (N <= 0 || B <= 0 || K < 0) return 0;
  return det_layout(N, B, K).total;
}

extern "C" int gsage_skipgram_bwd_det(const float *emb, int64_t N, int64_t D,
                                      const int64_t *targets, const int64_t *contexts,
                                      const int64_t *negatives, int64_t B, int64_t K,
                                      const float *coef, const float *grad_loss, float *grad_emb,
                                      void *ws, size_t ws_bytes, gsage_stream_t stream) {
  if (N <= 0 || D <= 0 || B <= 0 || K < 0 || !emb || !targets || !contexts || !coef ||
      !grad_loss || !grad_emb || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (K > 0 && !negatives) return GSAGE_ERR_INVALID_ARG;
  DetLayout L = det_layout(N, B, K);
  if (L.M >= (1ll << 31) - 1 || N >= (1ll << 31) - 2) return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < L.total) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  char *b = (char *)ws;
  int *key = (int *)(b + L.o_key), *tsrc = (int *)(b + L.o_src);
  float *tw = (float *)(b + L.o_tw);
  GsSortWs sw{(int *)(b + L.o_kA), (int *)(b + L.o_vA), (int *)(b + L.o_kB), (int *)(b + L.o_vB),
              (int *)(b + L.o_counts), (int *)(b + L.o_bsum)};
  int *rowptr = (int *)(b + L.o_rowptr);
  int *col = (int *)(b + L.o_col);
  float *w = (float *)(b + L.o_w);
  int *longs = (int *)(b + L.o_long), *nlong = (int *)(b + L.o_nlong);

  if (!gs_zero_async(nlong, 16, st)) return GSAGE_ERR_LAUNCH;
  hipLaunchKernelGGL(sg_terms_k, dim3(gs_grid(L.M, 256, 1 << 16)), dim3(256), 0, st, targets,
                     contexts, negatives, B, K, N, coef, grad_loss, key, tsrc, tw);
  GSAGE_CHECK_LAUNCH();
  int bits = 0;
  while ((1ll << bits) <= N) ++bits;
  const int *skey = nullptr;
  int rc = gs_radix_sort_vals2(key, tsrc, (const int *)tw, L.M, bits, col, (int *)w, sw,
                               (int *)(b + L.o_wA), (int *)(b + L.o_wB), st, &skey);
  if (rc) return rc;
  rc = gs_rowptr_from_sorted(skey, L.M, N, rowptr, st);   // row boundaries off the sorted keys
  if (rc) return rc;
  rc = gs_list_long_rows(rowptr, N, longs, L.long_cap, nlong, st);
  if (rc) return rc;
  return gsage_segment_gather_ew(rowptr, col, w, emb, N, D, grad_emb, longs, nlong, stream);
}

extern "C" int gsage_skipgram_bwd_rows_supported(int64_t D) { return (D == 64 || D == 128) ? 1 : 0; }

extern "C" size_t gsage_skipgram_bwd_rows_workspace_bytes(int64_t rows_cap, int64_t B, int64_t K) {
  if (rows_cap <= 0 || B <= 0 || K < 0) return 0;
  return rows_layout(rows_cap, B, K).total;
}

extern "C" int gsage_skipgram_bwd_rows(const float *emb, int64_t N, int64_t D,
                                       const int64_t *targets, const int64_t *contexts,
                                       const int64_t *negatives, int64_t B, int64_t K,
                                       const float *coef, const float *grad_loss, float *grad_emb,
                                       const uint32_t *live_mask, const int32_t *word_prefix,
                                       const int32_t *rows, const int32_t *n_rows, int64_t rows_cap,
                                       void *ws, size_t ws_bytes, gsage_stream_t stream) {
  if (N <= 0 || D <= 0 || B <= 0 || K < 0 || !emb || !targets || !contexts || !coef ||
      !grad_loss || !grad_emb || !live_mask || !word_prefix || !rows || !n_rows || !ws ||
      rows_cap <= 0)
    return GSAGE_ERR_INVALID_ARG;
  if (K > 0 && !negatives) return GSAGE_ERR_INVALID_ARG;
  if (!gsage_skipgram_bwd_rows_supported(D)) return GSAGE_ERR_UNSUPPORTED;
  if ((((uintptr_t)emb | (uintptr_t)grad_emb) & 15) != 0) return GSAGE_ERR_UNSUPPORTED;
  const RowsLayout L = rows_layout(rows_cap, B, K);
  if (L.M >= (1ll << 31) - 1 || N >= (1ll << 31) - 64 || rows_cap >= (1ll << 31) - 2)
    return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < L.total) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  char *b = (char *)ws;
  int *key = (int *)(b + L.o_key), *tsrc = (int *)(b + L.o_src);
  float *tw = (float *)(b + L.o_w);
  GsSortWs sw{(int *)(b + L.o_kA), (int *)(b + L.o_vA), (int *)(b + L.o_kB), (int *)(b + L.o_vB),
              (int *)(b + L.o_counts), (int *)(b + L.o_bsum)};
  int *rowptr = (int *)(b + L.o_rowptr);
  int *col = (int *)(b + L.o_col);
  float *w = (float *)(b + L.o_cw);
  // keys are places in the list, 0 .. rows_cap - 1, and the sentinel rows_cap
  hipLaunchKernelGGL(sg_terms_rows_k, dim3(gs_grid(L.M, 256, 1 << 16)), dim3(256), 0, st, targets,
                     contexts, negatives, B, K, N, coef, grad_loss, live_mask, word_prefix,
                     (int)rows_cap, key, tsrc, tw);
  GSAGE_CHECK_LAUNCH();
  int bits = 0;
  while ((1ll << bits) <= rows_cap) ++bits;
  const int *skey = nullptr;
  int rc = gs_radix_sort_vals2(key, tsrc, (const int *)tw, L.M, bits, col, (int *)w, sw,
                               (int *)(b + L.o_wA), (int *)(b + L.o_wB), st, &skey);
  if (rc) return rc;
  rc = gs_rowptr_from_sorted(skey, L.M, rows_cap, rowptr, st, n_rows);
  if (rc) return rc;
  const int G = D == 64 ? 4 : 2;
  const unsigned grid = (unsigned)((rows_cap + 4 * G - 1) / (4 * G));
  if (D == 64)
    hipLaunchKernelGGL((sg_rows_gather_k<16>), dim3(grid), dim3(256), 0, st, (const int *)rowptr,
                       (const int *)col, (const float *)w, emb, rows, n_rows, rows_cap, D, grad_emb);
  else
    hipLaunchKernelGGL((sg_rows_gather_k<32>), dim3(grid), dim3(256), 0, st, (const int *)rowptr,
                       (const int *)col, (const float *)w, emb, rows, n_rows, rows_cap, D, grad_emb);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}
