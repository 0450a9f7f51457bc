// Feeder kernels: uniform random walks over the out-neighbour CSR and the
// co-occurrence window rule that turns walks into (target, context) pairs.
//
// Reference: RandomWalkGenerator.generate_single_walk (src/walks/generator.py:119-146: start
// node, up to walk_length nodes, uniform choice among adj_list[current], stop at a dead end)
// and CooccurrencePairSampler.extract_pairs (src/walks/pair_sampler.py:56-86: for every
// position i, every j in [i-w, i+w] inside the walk, j != i, in that order).
// Integer work throughout; the pair rule is reproduced exactly, including the order.
#include "gsage_common.h"

namespace {

__device__ __forceinline__ uint32_t walk_hash(uint64_t seed, uint64_t walk, uint32_t step) {
  uint32_t h = (uint32_t)seed ^ ((uint32_t)walk * 0x9E3779B1u);
  h ^= (uint32_t)(seed >> 32) + (uint32_t)(walk >> 32) * 0x85EBCA77u + step * 0xC2B2AE3Du;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

// one thread per walk; walk w starts at node w / walks_per_node
__global__ void __launch_bounds__(256)
random_walks_k(const int *__restrict__ rowptr_t, const int *__restrict__ col_t, int64_t N,
               int L, int walks_per_node, uint64_t seed, int *__restrict__ walks,
               int *__restrict__ lengths) {
  const int64_t W = N * walks_per_node;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < W; w += stride) {
    int cur = (int)(w / walks_per_node);
    int *out = walks + w * L;
    out[0] = cur;
    int len = 1;
    for (int s = 1; s < L; ++s) {
      int p0 = rowptr_t[cur], deg = rowptr_t[cur + 1] - p0;
      if (deg == 0) break;                                   // dead end (:134-136)
      uint32_t r = walk_hash(seed, (uint64_t)w, (uint32_t)s);
      int pick = (int)(((uint64_t)r * (uint64_t)deg) >> 32);  // uniform in [0, deg)
      cur = col_t[p0 + pick];
      out[s] = cur;
      ++len;
    }
    for (int s = len; s < L; ++s) out[s] = -1;
    lengths[w] = len;
  }
}

// pairs contributed by position i of a walk of length len with window w
__device__ __forceinline__ int pos_pairs(int i, int len, int w) {
  int start = i - w > 0 ? i - w : 0;
  int end = i + w + 1 < len ? i + w + 1 : len;
  return end - start - 1;
}

__global__ void __launch_bounds__(256)
walk_pair_counts_k(const int *__restrict__ lengths, int64_t W, int window,
                   int64_t *__restrict__ counts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < W; w += stride) {
    int len = lengths[w];
    int64_t c = 0;
    for (int i = 0; i < len; ++i) c += pos_pairs(i, len, window);
    counts[w] = c;
  }
}

// pair number `id` in the reference's enumeration order -> (target, context)
__global__ void __launch_bounds__(256)
walk_pairs_gather_k(const int *__restrict__ walks, const int *__restrict__ lengths, int L,
                    const int64_t *__restrict__ offsets, int64_t W,
                    const int64_t *__restrict__ pair_ids, int64_t first_id, int64_t B, int window,
                    int64_t *__restrict__ targets, int64_t *__restrict__ contexts) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < B; q += stride) {
    const int64_t id = pair_ids ? pair_ids[q] : first_id + q;
    // walk = last w with offsets[w] <= id  (offsets has W+1 entries, non-decreasing)
    int64_t lo = 0, hi = W;
    while (hi - lo > 1) {
      int64_t mid = (lo + hi) >> 1;
      if (offsets[mid] <= id) lo = mid; else hi = mid;
    }
    const int64_t w = lo;
    int64_t rem = id - offsets[w];
    const int len = lengths[w];
    int i = 0;
    for (; i < len; ++i) {
      int c = pos_pairs(i, len, window);
      if (rem < c) break;
      rem -= c;
    }
    int start = i - window > 0 ? i - window : 0;
    int j = start + (int)rem;
    if (j >= i) ++j;                                          // the window skips j == i
    const int *wk = walks + w * L;
    targets[q] = wk[i];
    contexts[q] = wk[j];
  }
}

// ---- negatives by the alias method (DegreeBiasedNegativeSampler.sample,
// src/walks/negative_sampler.py:147-178: a uniform bin, a uniform threshold, `alias[bin]` where
// threshold >= prob_table[bin], else the bin).  One thread per draw; the random stream is a
// counter hash of (seed, draw number) -- three murmur3-finalised words: 64 bits pick the bin
// (bias N / 2^64), 24 bits make the threshold.  Mirrored on the host in graphsage_amd/rng.py.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}

__global__ void __launch_bounds__(256)
alias_sample_k(const float *__restrict__ prob_table, const int64_t *__restrict__ alias, int64_t N,
               int64_t count, uint64_t seed, uint64_t first_draw, int64_t *__restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < count; q += stride) {
    const uint64_t ctr = first_draw + (uint64_t)q;
    const uint32_t lo = (uint32_t)ctr, hi = (uint32_t)(ctr >> 32);
    const uint32_t a = fmix32(((uint32_t)seed ^ (lo * 0x9E3779B1u)) ^
                              ((uint32_t)(seed >> 32) + hi * 0x85EBCA77u));
    const uint32_t b = fmix32(a ^ 0x68E31DA4u ^ (lo * 0xB5297A4Du));
    const uint32_t c = fmix32((b + 0x1B56C4E9u + (uint32_t)(seed >> 32)) ^ hi);
    // bin = floor(x * N / 2^64), x = a:c, without a 128-bit product (N < 2^31)
    const uint64_t n = (uint64_t)N;
    const int64_t bin = (int64_t)(((uint64_t)a * n + (((uint64_t)c * n) >> 32)) >> 32);
    const float u = (float)(b >> 8) * (1.0f / 16777216.0f);
    out[q] = u >= prob_table[bin] ? alias[bin] : bin;
  }
}

}  // namespace

extern "C" int gsage_alias_sample(const float *prob_table, const int64_t *alias, int64_t N,
                                  int64_t count, uint64_t seed, uint64_t first_draw,
                                  int64_t *out, gsage_stream_t stream) {
  if (N <= 0 || count < 0 || !prob_table || !alias || !out) return GSAGE_ERR_INVALID_ARG;
  if (N >= (1ll << 31)) return GSAGE_ERR_UNSUPPORTED;
  if (count == 0) return GSAGE_OK;
  hipLaunchKernelGGL(alias_sample_k, dim3(gs_grid(count, 256 * 4, 256 * 32)), dim3(256), 0,
                     gs_stream(stream), prob_table, alias, N, count, seed, first_draw, out);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_random_walks(const int32_t *rowptr_t, const int32_t *col_t, int64_t N,
                                  int64_t walk_length, int64_t walks_per_node, uint64_t seed,
                                  int32_t *walks, int32_t *lengths, gsage_stream_t stream) {
  if (N < 0 || walk_length < 1 || walks_per_node < 1 || !rowptr_t || !walks || !lengths)
    return GSAGE_ERR_INVALID_ARG;
  if (walk_length > (1 << 20)) return GSAGE_ERR_UNSUPPORTED;
  if (N == 0) return GSAGE_OK;
  hipLaunchKernelGGL(random_walks_k, dim3(gs_grid(N * walks_per_node, 256, 256 * 32)), dim3(256), 0,
                     gs_stream(stream), rowptr_t, col_t, N, (int)walk_length, (int)walks_per_node,
                     seed, walks, lengths);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_walk_pair_counts(const int32_t *lengths, int64_t W, int64_t window,
                                      int64_t *counts, gsage_stream_t stream) {
  if (W < 0 || window < 0 || !lengths || !counts) return GSAGE_ERR_INVALID_ARG;
  if (W == 0) return GSAGE_OK;
  hipLaunchKernelGGL(walk_pair_counts_k, dim3(gs_grid(W, 256)), dim3(256), 0, gs_stream(stream),
                     lengths, W, (int)window, counts);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_walk_pairs_gather(const int32_t *walks, const int32_t *lengths,
                                       int64_t walk_length, const int64_t *offsets, int64_t W,
                                       const int64_t *pair_ids, int64_t first_id, int64_t B,
                                       int64_t window, int64_t *targets, int64_t *contexts,
                                       gsage_stream_t stream) {
  if (W <= 0 || B < 0 || window < 0 || !walks || !lengths || !offsets || !targets || !contexts)
    return GSAGE_ERR_INVALID_ARG;
  if (B == 0) return GSAGE_OK;
  hipLaunchKernelGGL(walk_pairs_gather_k, dim3(gs_grid(B, 256)), dim3(256), 0, gs_stream(stream),
                     walks, lengths, (int)walk_length, offsets, W, pair_ids, first_id, B,
                     (int)window, targets, contexts);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}
