// K1: CSR build for gfx950.  int64 [2,E] edge list -> int32 CSR in both orientations
// with stable (ascending edge id) order inside each row, plus inv_deg and hub-row lists.
//
// Method: validate+narrow to int32 -> degree histogram (integer atomics, exact) ->
// exclusive scan -> stable LSD radix sort of (key,value) pairs by key.  Each radix
// pass is histogram / scan / scatter; the scatter ranks items with wave ballots, so
// the order inside a bucket is the input order (tile, wave, round, lane) = stable.
// Everything is integer work: results are bit-exact and run-to-run identical.
#include "gsage_common.h"
#include "gsage_sort.h"

namespace {

constexpr int SCAN_T = 256;
constexpr int SCAN_IPT = 8;
constexpr int SCAN_CH = SCAN_T * SCAN_IPT;  // 2048 ints per block
static_assert(SCAN_CH == GS_SCAN_CH, "gsage_sort.h out of sync");

constexpr int RS_T = 256;                   // radix sort block
constexpr int RS_ROUNDS = 16;               // items per lane
constexpr int RS_TILE = RS_T * RS_ROUNDS;   // 4096 items per block
static_assert(RS_TILE == GS_RS_TILE, "gsage_sort.h out of sync");
constexpr int RS_WAVE_ITEMS = 64 * RS_ROUNDS;

constexpr int LONG_ROW_T = 512;             // rows longer than this take the hub pass

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = gs_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// ------------------------------------------------------------------ narrow + validate
// src32/dst32 <- edge_index rows; an edge with any index outside [0,N) gets key N on
// both sides (a sentinel row that sorts last and is never read) and raises the flag.
__global__ void narrow_validate_k(const int64_t *__restrict__ ei, int64_t E, int64_t N,
                                  int *__restrict__ src32, int *__restrict__ dst32,
                                  int *__restrict__ status) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad_any = false;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
    int64_t s = ei[e], d = ei[E + e];
    bool bad = (s < 0) | (s >= N) | (d < 0) | (d >= N);
    src32[e] = bad ? (int)N : (int)s;
    dst32[e] = bad ? (int)N : (int)d;
    bad_any |= bad;
  }
  if (bad_any) atomicOr(status, GSAGE_FLAG_INDEX_OUT_OF_RANGE);
}


__global__ void long_rows_k(const int *__restrict__ rowptr, int64_t N, int *__restrict__ list,
                            int64_t cap, int *__restrict__ count) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    if (rowptr[i + 1] - rowptr[i] > LONG_ROW_T) {
      int pos = atomicAdd(count, 1);
      if (pos < cap) list[pos] = (int)i;
    }
  }
}

// ------------------------------------------------------------------------------ scan
__global__ void scan_reduce_k(const int *__restrict__ in, int64_t n, int *__restrict__ bsum) {
  __shared__ int sh[SCAN_T / 64];
  int64_t base = (int64_t)blockIdx.x * SCAN_CH;
  int s = 0;
  for (int i = threadIdx.x; i < SCAN_CH; i += SCAN_T) {
    int64_t g = base + i;
    if (g < n) s += in[g];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (gs_lane() == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < SCAN_T / 64; ++w) t += sh[w];
    bsum[blockIdx.x] = t;
  }
}

// one block of 1024 threads: exclusive scan of bsum[nb] in place
__global__ void scan_spine_k(int *__restrict__ bsum, int64_t nb) {
  __shared__ int wsum[16];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  int carry = 0;
  for (int64_t base = 0; base < nb; base += 1024) {
    int64_t i = base + threadIdx.x;
    int v = i < nb ? bsum[i] : 0;
    int inc = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      int x = wsum[w];
      if (w < wave) woff += x;
      total += x;
    }
    if (i < nb) bsum[i] = carry + woff + inc - v;
    carry += total;
    __syncthreads();
  }
}

__global__ void scan_apply_k(const int *__restrict__ in, int64_t n,
                             const int *__restrict__ bsum_excl, int *__restrict__ out) {
  __shared__ int ch[SCAN_CH];
  __shared__ int wsum[SCAN_T / 64];
  const int t = threadIdx.x, lane = gs_lane(), wave = t >> 6;
  int64_t base = (int64_t)blockIdx.x * SCAN_CH;
  for (int i = t; i < SCAN_CH; i += SCAN_T) {
    int64_t g = base + i;
    ch[i] = g < n ? in[g] : 0;
  }
  __syncthreads();
  int loc[SCAN_IPT];
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) {
    loc[j] = ch[t * SCAN_IPT + j];
    s += loc[j];
  }
  int inc = wave_incl_scan(s);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  int run = bsum_excl[blockIdx.x] + woff + inc - s;
#pragma unroll
  for (int j = 0; j < SCAN_IPT; ++j) {
    ch[t * SCAN_IPT + j] = run;
    run += loc[j];
  }
  __syncthreads();
  for (int i = t; i < SCAN_CH; i += SCAN_T) {
    int64_t g = base + i;
    if (g < n) out[g] = ch[i];
  }
}

}  // namespace

// exclusive scan of in[0..n) into out[0..n) (in may equal out); bsum: ceil(n/SCAN_CH) ints
int gs_exclusive_scan(const int *in, int64_t n, int *out, int *bsum, hipStream_t st) {
  if (n <= 0) return GSAGE_OK;
  int64_t nb = (n + SCAN_CH - 1) / SCAN_CH;
  hipLaunchKernelGGL(scan_reduce_k, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, bsum);
  hipLaunchKernelGGL(scan_spine_k, dim3(1), dim3(1024), 0, st, bsum, nb);
  hipLaunchKernelGGL(scan_apply_k, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, bsum, out);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

int gs_list_long_rows(const int *rowptr, int64_t N, int *list, int64_t cap, int *count,
                      hipStream_t st) {
  if (N <= 0) return GSAGE_OK;
  hipLaunchKernelGGL(long_rows_k, dim3(gs_grid(N, 256)), dim3(256), 0, st, rowptr, N, list, cap,
                     count);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

namespace {

// ------------------------------------------------------------------------ radix pass
__global__ void radix_hist_k(const int *__restrict__ keys, int64_t n, int shift, int mask,
                             int *__restrict__ counts, int64_t ntiles) {
  __shared__ int h[256];
  const int t = threadIdx.x;
  h[t] = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ROUNDS; ++r) {
    int64_t idx = base + (int64_t)r * RS_T + t;
    if (idx < n) atomicAdd(&h[(keys[idx] >> shift) & mask], 1);
  }
  __syncthreads();
  if (t <= mask) counts[(int64_t)t * ntiles + blockIdx.x] = h[t];
}

// TWO: a second 32-bit payload travels with every item (the loss backward sorts (row, source,
// weight) triples, so no permutation has to be applied afterwards).
// The tile's items are first put in digit order inside LDS (stable: wave, round, lane), then
// copied out by consecutive threads: an item's neighbours in LDS are its neighbours in memory as
// long as they share its digit (16 items = 64 bytes on average at 256 digits per 4 096-item tile),
// where a direct scatter writes 4 bytes at a time to 64 places per instruction (round 3: 224 ->
// see DESIGN.md, K1).
template <bool TWO>
__global__ void __launch_bounds__(RS_T)
radix_scatter_k(const int *__restrict__ kin, const int *__restrict__ vin, int64_t n,
                int shift, int mask, int width, const int *__restrict__ offs,
                int64_t ntiles, int *__restrict__ kout, int *__restrict__ vout,
                const int *__restrict__ v2in, int *__restrict__ v2out) {
  __shared__ int whist[RS_T / 64][256];     // per wave and digit: count, then next place in the LDS image
  __shared__ int lstart[256];               // first place of a digit in the LDS image
  __shared__ int gbase[256];                // where the tile's first item of a digit goes in memory
  __shared__ int wsum[RS_T / 64];
  __shared__ int sk[RS_TILE], sv[RS_TILE], sv2[TWO ? RS_TILE : 1];
  const int t = threadIdx.x, lane = gs_lane(), w = t >> 6;
  for (int i = t; i < (RS_T / 64) * 256; i += RS_T) (&whist[0][0])[i] = 0;
  __syncthreads();

  const int64_t tbase = (int64_t)blockIdx.x * RS_TILE;
  const int64_t wbase = tbase + (int64_t)w * RS_WAVE_ITEMS;
  int k[RS_ROUNDS], v[RS_ROUNDS], v2[TWO ? RS_ROUNDS : 1];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    int64_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    k[r] = valid ? kin[idx] : 0;
    v[r] = valid ? vin[idx] : 0;
    if (TWO) v2[r] = valid ? v2in[idx] : 0;
  }
  // phase 1: per-wave digit histogram
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    int64_t idx = wbase + r * 64 + lane;
    if (idx < n) atomicAdd(&whist[w][(k[r] >> shift) & mask], 1);
  }
  __syncthreads();
  // per digit: the waves in order, then the digits in order (block-wide exclusive scan of the totals)
  {
    int tot = 0;
#pragma unroll
    for (int ww = 0; ww < RS_T / 64; ++ww) {
      int c = whist[ww][t];
      whist[ww][t] = tot;
      tot += c;
    }
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int before = 0;
    for (int ww = 0; ww < w; ++ww) before += wsum[ww];
    const int ls = before + inc - tot;
    lstart[t] = ls;
    gbase[t] = t <= mask ? offs[(int64_t)t * ntiles + blockIdx.x] : 0;
#pragma unroll
    for (int ww = 0; ww < RS_T / 64; ++ww) whist[ww][t] += ls;
  }
  __syncthreads();
  // phase 2: stable ranks inside the wave, round by round, into the LDS image
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    int64_t idx = wbase + r * 64 + lane;
    bool valid = idx < n;
    int d = (k[r] >> shift) & mask;
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < width; ++b) {
      bool bit = (d >> b) & 1;
      unsigned long long bb = __ballot(bit);
      m &= bit ? bb : ~bb;
    }
    int rank = __popcll(m & ((1ull << lane) - 1ull));
    int leader = __ffsll((long long)m) - 1;
    int basev = 0;
    if (valid && lane == leader) {
      basev = whist[w][d];
      whist[w][d] = basev + __popcll(m);
    }
    basev = __shfl(basev, leader & 63, 64);
    if (valid) {
      int pos = basev + rank;
      sk[pos] = k[r];
      sv[pos] = v[r];
      if (TWO) sv2[pos] = v2[r];
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // phase 3: out, consecutive threads to consecutive places of a digit's run
  int64_t left = n - tbase;
  const int cnt = left < RS_TILE ? (int)left : RS_TILE;
#pragma unroll 4
  for (int i = t; i < cnt; i += RS_T) {
    const int key = sk[i];
    const int d = (key >> shift) & mask;
    const int pos = gbase[d] + (i - lstart[d]);
    if (kout) kout[pos] = key;
    vout[pos] = sv[i];
    if (TWO) v2out[pos] = sv2[i];
  }
}

}  // namespace

// stable sort of (key,val[,val2]) by key in [0, 2^bits); the sorted values land in vout_final
// (and v2out_final).  vals2 == nullptr: one payload.
int gs_radix_sort_vals2(const int *keys, const int *vals, const int *vals2, int64_t n, int bits,
                        int *vout_final, int *v2out_final, const GsSortWs &ws, int *v2A, int *v2B,
                        hipStream_t st, const int **sorted_keys) {
  if (sorted_keys) *sorted_keys = keys;
  if (n <= 0) return GSAGE_OK;
  if (bits < 1) bits = 1;
  int passes = (bits + 7) / 8;
  int width = (bits + passes - 1) / passes;
  int64_t ntiles = (n + RS_TILE - 1) / RS_TILE;
  const int *kin = keys, *vin = vals, *v2in = vals2;
  for (int p = 0; p < passes; ++p) {
    int shift = p * width;
    int mask = (1 << width) - 1;
    bool last = (p == passes - 1);
    int *kout = (last && !sorted_keys) ? nullptr : ((p & 1) ? ws.kB : ws.kA);
    int *vout = last ? vout_final : ((p & 1) ? ws.vB : ws.vA);
    int *v2out = last ? v2out_final : ((p & 1) ? v2B : v2A);
    hipLaunchKernelGGL(radix_hist_k, dim3((unsigned)ntiles), dim3(RS_T), 0, st, kin, n, shift,
                       mask, ws.counts, ntiles);
    int64_t ncounts = (int64_t)(mask + 1) * ntiles;
    int rc = gs_exclusive_scan(ws.counts, ncounts, ws.counts, ws.bsum, st);
    if (rc) return rc;
    if (vals2)
      hipLaunchKernelGGL(radix_scatter_k<true>, dim3((unsigned)ntiles), dim3(RS_T), 0, st, kin, vin, n,
                         shift, mask, width, ws.counts, ntiles, kout, vout, v2in, v2out);
    else
      hipLaunchKernelGGL(radix_scatter_k<false>, dim3((unsigned)ntiles), dim3(RS_T), 0, st, kin, vin, n,
                         shift, mask, width, ws.counts, ntiles, kout, vout, (const int *)nullptr,
                         (int *)nullptr);
    GSAGE_CHECK_LAUNCH();
    kin = kout;
    vin = vout;
    v2in = v2out;
  }
  if (sorted_keys) *sorted_keys = kin;
  return GSAGE_OK;
}

int gs_radix_sort_vals(const int *keys, const int *vals, int64_t n, int bits, int *vout_final,
                       const GsSortWs &ws, hipStream_t st, const int **sorted_keys) {
  return gs_radix_sort_vals2(keys, vals, nullptr, n, bits, vout_final, nullptr, ws, nullptr, nullptr,
                             st, sorted_keys);
}

namespace {
// position p in [0, n] starts rows (key[p-1], key[p]] (key[-1] = -1, key[n] = nrows); the lanes
// of a wave fill each such range together, so a long run of empty rows is written coalesced
__global__ void __launch_bounds__(256)
rowptr_from_sorted_k(const int *__restrict__ skey, int64_t n, int64_t nrows, int *__restrict__ rp,
                     const int *__restrict__ nrows_dev) {
  const int lane = gs_lane();
  // the row count may live on the device (a per-batch CSR over a row list whose length the host
  // does not know): rows beyond it are not wanted, so the end of the keys closes row *nrows_dev
  // instead of filling rowptr up to the host's bound
  if (nrows_dev && *nrows_dev < nrows) nrows = *nrows_dev;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t base = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); base <= n; base += stride) {
    const int64_t p = base + lane;
    int prev = 0, cur = 0;
    if (p <= n) {
      prev = p > 0 ? skey[p - 1] : -1;
      cur = p < n ? skey[p] : (int)nrows;
      if (cur < prev) cur = prev;      // (a sentinel key above a device-side row count: no new row)
    }
    // most boundaries start ONE row (dense keys: the per-batch CSR of the loss backward has a
    // boundary every 2.4 positions); those lanes write their few entries themselves and only the
    // long runs of empty rows are filled by the whole wave
    const int gap = cur - prev;
    if (gap > 0 && gap <= 4)
      for (int k = prev + 1; k <= cur; ++k) rp[k] = (int)p;
    uint64_t m = __ballot(gap > 4);
    while (m) {
      const int l = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      const int lo = __shfl(prev, l, 64) + 1, hi = __shfl(cur, l, 64);
      const int v = (int)(base + l);
      for (int k = lo + lane; k <= hi; k += 64) rp[k] = v;
    }
  }
}

__global__ void __launch_bounds__(256)
inv_deg_from_rowptr_k(const int *__restrict__ rp, int64_t N, float *__restrict__ inv_deg) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
    int d = rp[i + 1] - rp[i];
    inv_deg[i] = 1.0f / (float)(d > 1 ? d : 1);
  }
}
}  // namespace

int gs_rowptr_from_sorted(const int *sorted_keys, int64_t n, int64_t nrows, int *rowptr,
                          hipStream_t st, const int *nrows_dev) {
  hipLaunchKernelGGL(rowptr_from_sorted_k, dim3(gs_grid(n + 1, 256, 1 << 20)), dim3(256), 0, st,
                     sorted_keys, n, nrows, rowptr, nrows_dev);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

namespace {

struct CsrWsLayout {
  size_t total;
  size_t o_src, o_dst, o_kA, o_vA, o_kB, o_vB, o_counts, o_bsum;
};

CsrWsLayout csr_layout(int64_t E, int64_t N) {
  CsrWsLayout L;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += gs_align_up(bytes ? bytes : 4, 256);
    return o;
  };
  size_t e4 = (size_t)(E > 0 ? E : 1) * 4;
  int64_t ntiles = (E + RS_TILE - 1) / RS_TILE;
  if (ntiles < 1) ntiles = 1;
  int64_t ncounts = 256 * ntiles;
  int64_t scan_n = ncounts > (N + 2) ? ncounts : (N + 2);
  L.o_src = take(e4);
  L.o_dst = take(e4);
  L.o_kA = take(e4);
  L.o_vA = take(e4);
  L.o_kB = take(e4);
  L.o_vB = take(e4);
  L.o_counts = take((size_t)ncounts * 4);
  L.o_bsum = take((size_t)((scan_n + SCAN_CH - 1) / SCAN_CH + 1) * 4);
  L.total = off;
  return L;
}

}  // namespace

extern "C" int gsage_long_row_threshold(void) { return LONG_ROW_T; }

extern "C" size_t gsage_csr_workspace_bytes(int64_t E, int64_t N) {
  if (E < 0 || N < 0) return 0;
  return csr_layout(E, N).total;
}

extern "C" int gsage_csr_build(const int64_t *edge_index, int64_t E, int64_t N, int32_t *rowptr,
                               int32_t *col, int32_t *rowptr_t, int32_t *col_t, float *inv_deg,
                               int32_t *long_rows, int32_t *long_rows_t, int64_t long_cap,
                               int32_t *n_long, int32_t *status, void *ws, size_t ws_bytes,
                               gsage_stream_t stream) {
  if (E < 0 || N < 0 || !rowptr || !rowptr_t || !inv_deg || !status || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (E > 0 && (!edge_index || !col || !col_t)) return GSAGE_ERR_INVALID_ARG;
  if (E >= (1ll << 31) - 1 || N >= (1ll << 31) - 2) return GSAGE_ERR_UNSUPPORTED;
  // at most E / threshold rows can be longer than the threshold: a smaller list could be
  // truncated, and the gather kernels would then skip rows nobody owns
  if ((long_rows || long_rows_t) && (!n_long || long_cap < E / LONG_ROW_T + 1))
    return GSAGE_ERR_INVALID_ARG;
  CsrWsLayout L = csr_layout(E, N);
  if (ws_bytes < L.total) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  char *b = (char *)ws;
  int *src32 = (int *)(b + L.o_src), *dst32 = (int *)(b + L.o_dst);
  GsSortWs sw{(int *)(b + L.o_kA), (int *)(b + L.o_vA), (int *)(b + L.o_kB),
            (int *)(b + L.o_vB), (int *)(b + L.o_counts), (int *)(b + L.o_bsum)};

  if (E > 0) {
    hipLaunchKernelGGL(narrow_validate_k, dim3(gs_grid(E, 256)), dim3(256), 0, st, edge_index, E,
                       N, src32, dst32, status);
    GSAGE_CHECK_LAUNCH();
  }
  int bits = 0;
  while ((1ll << bits) <= N) ++bits;  // bits needed to hold the sentinel key N

  if (n_long && !gs_zero_async(n_long, 2 * sizeof(int32_t), st))
    return GSAGE_ERR_LAUNCH;

  for (int orient = 0; orient < 2; ++orient) {
    const int *key = orient == 0 ? dst32 : src32;
    const int *val = orient == 0 ? src32 : dst32;
    int32_t *rp = orient == 0 ? rowptr : rowptr_t;
    int32_t *cl = orient == 0 ? col : col_t;
    // sort first, then read the row boundaries off the sorted keys: no degree histogram (200 M
    // integer atomics cost 9 ms per orientation at C4, half of the whole build)
    const int *skey = nullptr;
    int rc = gs_radix_sort_vals(key, val, E, bits, cl, sw, st, &skey);
    if (rc) return rc;
    rc = gs_rowptr_from_sorted(skey, E, N, rp, st);
    if (rc) return rc;
    if (orient == 0 && N > 0) {
      hipLaunchKernelGGL(inv_deg_from_rowptr_k, dim3(gs_grid(N, 256)), dim3(256), 0, st, rp, N,
                         inv_deg);
      GSAGE_CHECK_LAUNCH();
    }
    int32_t *lr = orient == 0 ? long_rows : long_rows_t;
    if (lr && n_long && N > 0 && long_cap > 0) {
      rc = gs_list_long_rows(rp, N, lr, long_cap, n_long + orient, st);
      if (rc) return rc;
    }
  }
  return GSAGE_OK;
}
