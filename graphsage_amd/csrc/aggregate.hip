// K2 / K7: segment gather over a CSR (SAGEConv mean aggregation and its transposed
// backward) for gfx950.  HBM-bound: the only arithmetic is one add per gathered float.
//
//   out[i,:] = add_to[i,:] + s_i * sum_{p in row i} w_p * x[col[p],:]
//
// Layout: one wavefront per destination row.  A feature row of F floats is covered by
// LPR = F/4 lanes with one 16-byte load each, so one wave-instruction fetches
// G = 64/LPR whole neighbour rows (F=64: 4 rows of 256 B).  The row's neighbour ids
// are fetched 64 at a time with one coalesced load and handed to the lanes with
// ds_bpermute (the LDS crossbar; no LDS memory is touched).  Sums are kept in
// registers and combined across the G lane groups by xor-shuffles in a fixed order,
// so results are bitwise reproducible.  Rows longer than the hub threshold are taken by
// the first HUB_BLOCKS workgroups of the same launch (dispatched first, so the long rows
// overlap the rest of the grid): 4 waves share a hub row in interleaved 64-neighbour
// chunks and combine through LDS in wave order.
#include "gsage_common.h"

namespace {

constexpr int AGG_BLOCK = 256;         // 4 waves, one row per wave per iteration
#ifndef GATHER_NB
#define GATHER_NB 4
#endif
#ifndef MASK_NB
#define MASK_NB 4
#endif
constexpr int HUB_BLOCKS = 256;        // leading workgroups that walk the hub-row list

template <int LPR>
__device__ __forceinline__ float4 group_reduce(float4 a) {
  // sum the G = 64/LPR lane groups; afterwards every lane holds the total of its column
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    a.x += __shfl_xor(a.x, o, 64);
    a.y += __shfl_xor(a.y, o, 64);
    a.z += __shfl_xor(a.z, o, 64);
    a.w += __shfl_xor(a.w, o, 64);
  }
  return a;
}

// accumulate neighbours p in [p0, p1) of one row, visiting them in steps of
// `stride_groups` neighbour slots (this wave owns every slot; hub waves share).
//
// MASKED: `mask` holds one bit per row of x, clear = that row is known to be all zero.  The
// chunk's 64 neighbour ids are tested with one (L2-resident) word load per lane and compacted
// with a ballot + ds_permute, so only the live neighbours' rows are fetched and the load slots
// stay full.  Skipping adds of +-0 changes no sum; compaction changes which lane group adds which
// neighbour, i.e. the rounding order, still fixed for a given mask.
template <int LPR, int WMODE, bool MASKED>
__device__ __forceinline__ float4 gather_range(const int *__restrict__ col,
                                               const float *__restrict__ x,
                                               const float *__restrict__ ew,
                                               const uint32_t *__restrict__ mask, int64_t F,
                                               int p0, int p1) {
  constexpr int G = 64 / LPR;
  constexpr int NB = GATHER_NB;
  static_assert(!(MASKED && WMODE == 2), "per-entry weights do not survive the compaction");
  const int lane = gs_lane();
  const int slot = lane / LPR, sub = lane % LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int pc = p0; pc < p1; pc += 64) {
    int pl = pc + lane;
    int cv = pl < p1 ? col[pl] : 0;
    int cnt = p1 - pc;
    if (cnt > 64) cnt = 64;
    if (MASKED) {
      const bool live = pl < p1 && ((mask[cv >> 5] >> (cv & 31)) & 1u);
      const unsigned long long b = __ballot(live);
      const int nlive = __popcll(b);
      const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32),
                                                  __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
      // a full permutation: live ids to the front in order, the rest behind them
      const int pos = live ? before : nlive + (lane - before);
      cv = __builtin_amdgcn_ds_permute(pos << 2, cv);
      cnt = nlive;
    }
    int steps = (cnt + G - 1) / G;
    // NB independent 16-byte loads in flight per lane; the last batch is predicated rather than
    // peeled, so a short row (most rows of a power-law graph) still has all its loads in flight.
    // (plain ifs: a ?: between the loaded lvalue and a zero lvalue becomes a load through a
    // selected pointer, i.e. flat loads and a zero vector in scratch)
    for (int s = 0; s < steps; s += NB) {
      float4 r[NB];
      float w[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int j = (s + k) * G + slot;
        const int c = __shfl(cv, j, 64);
        r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        w[k] = 0.f;
        if (j < cnt) {
          r[k] = *(const float4 *)(x + (int64_t)c * F + sub * 4);
          if (WMODE == 1) w[k] = ew[c];        // weight of the source node
          if (WMODE == 2) w[k] = ew[pc + j];   // weight of the edge (same order as col)
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (WMODE != 0) {
          acc.x += r[k].x * w[k]; acc.y += r[k].y * w[k];
          acc.z += r[k].z * w[k]; acc.w += r[k].w * w[k];
        } else {
          acc.x += r[k].x; acc.y += r[k].y; acc.z += r[k].z; acc.w += r[k].w;
        }
      }
    }
  }
  return acc;
}

// hub rows: one workgroup (4 waves) per row; waves take interleaved 64-neighbour chunks,
// partial sums are combined through LDS in wave order (fixed => reproducible).
template <int LPR, int WMODE, bool MASKED>
__device__ __forceinline__ void hub_rows(const int *__restrict__ rowptr, const int *__restrict__ col,
                                         const float *__restrict__ row_scale,
                                         const float *__restrict__ ew,
                                         const uint32_t *__restrict__ mask,
                                         const float *__restrict__ x,
                                         const float *add_to, const uint32_t *__restrict__ amask,
                                         int64_t F, float *out,
                                         const int *__restrict__ long_rows,
                                         const int *__restrict__ n_long, int hub_blocks) {
  constexpr int NW = AGG_BLOCK / 64;
  __shared__ float4 part[NW][LPR];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  const int slot = lane / LPR, sub = lane % LPR;
  const int nl = *n_long;
  for (int li = blockIdx.x; li < nl; li += hub_blocks) {
    int64_t row = long_rows[li];
    int p0 = rowptr[row], p1 = rowptr[row + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pc = p0 + wave * 64; pc < p1; pc += NW * 64) {
      int pe = pc + 64 < p1 ? pc + 64 : p1;
      float4 r = gather_range<LPR, WMODE, MASKED>(col, x, ew, mask, F, pc, pe);
      acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
    }
    acc = group_reduce<LPR>(acc);
    if (slot == 0) part[wave][sub] = acc;
    __syncthreads();
    if (wave == 0 && slot == 0) {
      float4 t = part[0][sub];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        float4 q = part[w][sub];
        t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
      }
      float s = row_scale ? row_scale[row] : 1.f;
      float4 o = make_float4(t.x * s, t.y * s, t.z * s, t.w * s);
      int64_t off = row * F + sub * 4;
      if (add_to && (!amask || ((amask[row >> 5] >> (row & 31)) & 1u))) {
        float4 a = *(const float4 *)(add_to + off);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      *(float4 *)(out + off) = o;
    }
    __syncthreads();
  }
}

// Masked gather of the G (2 or 4) consecutive rows of one wave as ONE flat list: with about half
// of the neighbours masked out the rows are short and a row-after-row walk is bound by its chain
// of dependent loads (rowptr -> col -> mask -> rows -> add_to).  Here the wave reads the G rows'
// col entries 128 at a time regardless of the row boundaries, tests them against the mask and
// writes the live ones, each with the number of the row it belongs to, into a wave-private LDS
// list; the lane groups then walk that list with all their load slots full and add every fetched
// neighbour to the accumulator of its row.  The add_to rows are requested before anything else.
// Returns false, having done nothing, when one of the rows belongs to the hub pass.
template <int LPR, int WMODE>
__device__ __forceinline__ bool masked_flat_rows(const int *__restrict__ rowptr,
                                                 const int *__restrict__ col,
                                                 const float *__restrict__ x,
                                                 const float *__restrict__ ew,
                                                 const uint32_t *__restrict__ mask,
                                                 const float *add_to,
                                                 const uint32_t *__restrict__ amask, int64_t N,
                                                 int64_t F, int64_t rg, int long_t,
                                                 int2 *nbr /* [128] */, float4 &keep, float4 &base) {
  constexpr int G = 64 / LPR;
  constexpr int NB = MASK_NB;
  static_assert(G == 2 || G == 4, "flat walk keeps one accumulator per row");
  const int lane = gs_lane();
  const int slot = lane / LPR, sub = lane % LPR;
  int b[G + 1];
#pragma unroll
  for (int i = 0; i <= G; ++i) b[i] = rowptr[rg + i < N ? rg + i : N];
  // no branch around any load of this function: a load inside a conditional block is followed
  // by a wait for it (the copy into the merged registers), which would put the loads in series.
  // Out-of-range lanes read a clamped address instead and drop the value with a select.
  {
    // amask: a clear bit says that row of add_to is zero BY PROMISE (it may hold anything):
    // such a row is neither fetched (the lane reads row 0 of x, a cached line) nor added
    const int64_t rowc = rg + slot < N ? rg + slot : N - 1;
    const bool have = add_to && (!amask || ((amask[rowc >> 5] >> (rowc & 31)) & 1u));
    const float *bp = have ? add_to + rowc * F + sub * 4 : x + sub * 4;
    const float4 v = *(const float4 *)bp;
    base = have ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (long_t > 0) {
#pragma unroll
    for (int i = 0; i < G; ++i)
      if (b[i + 1] - b[i] > long_t) return false;
  }
  float4 acc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int pend = b[G];
  for (int pc = b[0]; pc < pend; pc += 128) {
    const int pl0 = pc + lane, pl1 = pc + 64 + lane;
    const int c0 = col[pl0 < pend ? pl0 : pend - 1], c1 = col[pl1 < pend ? pl1 : pend - 1];
    const uint32_t m0 = mask[c0 >> 5], m1 = mask[c1 >> 5];
    int r0 = 0, r1 = 0;
#pragma unroll
    for (int i = 1; i < G; ++i) {
      r0 += pl0 >= b[i] ? 1 : 0;
      r1 += pl1 >= b[i] ? 1 : 0;
    }
    const bool live0 = pl0 < pend && ((m0 >> (c0 & 31)) & 1u);
    const bool live1 = pl1 < pend && ((m1 >> (c1 & 31)) & 1u);
    const unsigned long long bal0 = __ballot(live0), bal1 = __ballot(live1);
    const int cnt0 = __popcll(bal0), cnt = cnt0 + __popcll(bal1);
    if (live0)
      nbr[__builtin_amdgcn_mbcnt_hi((unsigned)(bal0 >> 32),
                                    __builtin_amdgcn_mbcnt_lo((unsigned)bal0, 0))] = make_int2(c0, r0);
    if (live1)
      nbr[cnt0 + __builtin_amdgcn_mbcnt_hi((unsigned)(bal1 >> 32),
                                           __builtin_amdgcn_mbcnt_lo((unsigned)bal1, 0))] = make_int2(c1, r1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int steps = (cnt + G - 1) / G;
    for (int s = 0; s < steps; s += NB) {
      float4 r[NB];
      float w[NB];
      int2 e[NB];
      // the list entries first (LDS latency paid once), then the row loads back to back
#pragma unroll
      for (int k = 0; k < NB; ++k) e[k] = nbr[((s + k) * G + slot) & 127];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        // slots past the end of the list fetch row 0 (cached after its first use) and get the
        // row number -1, which no accumulator answers to
        const bool valid = (s + k) * G + slot < cnt;
        const int c = valid ? e[k].x : 0;
        e[k].y = valid ? e[k].y : -1;
        r[k] = *(const float4 *)(x + (int64_t)c * F + sub * 4);
        w[k] = WMODE == 1 ? ew[c] : 1.f;
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const bool hit = e[k].y == i;   // select the product, not the weight: 0 * inf is NaN
          acc[i].x += hit ? r[k].x * w[k] : 0.f; acc[i].y += hit ? r[k].y * w[k] : 0.f;
          acc[i].z += hit ? r[k].z * w[k] : 0.f; acc[i].w += hit ? r[k].w * w[k] : 0.f;
        }
      }
    }
    // the next round overwrites the list: every lane must have read its entries
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const float4 t = group_reduce<LPR>(acc[i]);
    if (slot == i) keep = t;
  }
  return true;
}

// ---------------------------------------------------------------- vector path, F = 4*LPR
#ifndef MASK_WAVES
#define MASK_WAVES 6
#endif
template <int LPR, int WMODE, bool MASKED>
__global__ void __launch_bounds__(AGG_BLOCK, MASKED ? MASK_WAVES : 1)
seg_gather_vec_k(const int *__restrict__ rowptr, const int *__restrict__ col,
                 const float *__restrict__ row_scale, const float *__restrict__ ew,
                 const uint32_t *__restrict__ mask, const float *__restrict__ x, const float *add_to,
                 const uint32_t *__restrict__ amask, int64_t N, int64_t F,
                 float *out, int long_t, const int *__restrict__ long_rows,
                 const int *__restrict__ n_long, int hub_blocks) {
  if ((int)blockIdx.x < hub_blocks) {
    hub_rows<LPR, WMODE, MASKED>(rowptr, col, row_scale, ew, mask, x, add_to, amask, F, out,
                                 long_rows, n_long, hub_blocks);
    return;
  }
  constexpr int G = 64 / LPR;
  const int lane = gs_lane();
  const int slot = lane / LPR, sub = lane % LPR;
  const int64_t wave0 = (int64_t)(blockIdx.x - hub_blocks) * (AGG_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)(gridDim.x - hub_blocks) * (AGG_BLOCK / 64);
  // a wave takes G consecutive rows per iteration and stores them with ONE full-wave
  // 1-KiB store (lane group g keeps row g's sum)
  for (int64_t rg = wave0 * G; rg < N; rg += nwaves * G) {
    float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
    bool mine = false;
    if constexpr (MASKED && WMODE != 2 && (G == 2 || G == 4)) {
      __shared__ int2 nbr[AGG_BLOCK / 64][128];
      float4 base;
      if (masked_flat_rows<LPR, WMODE>(rowptr, col, x, ew, mask, add_to, amask, N, F, rg, long_t,
                                       nbr[threadIdx.x >> 6], keep, base)) {
        if (rg + slot < N) {
          const int64_t row = rg + slot;
          const float s = row_scale ? row_scale[row] : 1.f;
          *(float4 *)(out + row * F + sub * 4) =
              make_float4(base.x + keep.x * s, base.y + keep.y * s, base.z + keep.z * s,
                          base.w + keep.w * s);
        }
        continue;
      }
    }
#pragma unroll 1
    for (int i = 0; i < G; ++i) {
      int64_t row = rg + i;
      if (row >= N) break;
      int p0 = rowptr[row], p1 = rowptr[row + 1];
      if (long_t > 0 && p1 - p0 > long_t) continue;  // hub pass owns this row
      float4 acc = gather_range<LPR, WMODE, MASKED>(col, x, ew, mask, F, p0, p1);
      acc = group_reduce<LPR>(acc);
      if (slot == i) {
        keep = acc;
        mine = true;
      }
    }
    if (mine) {
      int64_t row = rg + slot;
      float s = row_scale ? row_scale[row] : 1.f;
      float4 o = make_float4(keep.x * s, keep.y * s, keep.z * s, keep.w * s);
      int64_t off = row * F + sub * 4;
      if (add_to && (!amask || ((amask[row >> 5] >> (row & 31)) & 1u))) {
        float4 a = *(const float4 *)(add_to + off);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      *(float4 *)(out + off) = o;
    }
  }
}

// ------------------------------------------------------- scalar path, any F (e.g. F = 7)
// P = lanes per neighbour slot (power of two >= min(F,64)); each lane owns feature
// columns sub, sub+P, ... of its slot's neighbour.  Up to 4 columns per lane are kept
// in registers per sweep (F <= 4*P per sweep; larger F sweeps the row again).
template <int WMODE>
__global__ void __launch_bounds__(AGG_BLOCK)
seg_gather_scalar_k(const int *__restrict__ rowptr, const int *__restrict__ col,
                    const float *__restrict__ row_scale, const float *__restrict__ ew,
                    const float *__restrict__ x, const float *add_to, int64_t N, int64_t F,
                    float *out, int P) {
  const int lane = gs_lane();
  const int G = 64 / P;
  const int slot = lane / P, sub = lane % P;
  const int64_t wave0 = (int64_t)blockIdx.x * (AGG_BLOCK / 64) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (AGG_BLOCK / 64);
  for (int64_t row = wave0; row < N; row += nwaves) {
    int p0 = rowptr[row], p1 = rowptr[row + 1];
    float s = row_scale ? row_scale[row] : 1.f;
    for (int64_t f0 = 0; f0 < F; f0 += P) {
      int64_t f = f0 + sub;
      float acc = 0.f;
      for (int pc = p0; pc < p1; pc += 64) {
        int pl = pc + lane;
        int cv = pl < p1 ? col[pl] : 0;
        int cnt = p1 - pc;
        if (cnt > 64) cnt = 64;
        int steps = (cnt + G - 1) / G;
        for (int st = 0; st < steps; ++st) {
          int j = st * G + slot;
          int c = __shfl(cv, j, 64);
          if (j < cnt && f < F) {
            float w = WMODE == 1 ? ew[c] : WMODE == 2 ? ew[pc + j] : 1.f;
            acc += x[(int64_t)c * F + f] * w;
          }
        }
      }
      for (int o = P; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
      if (slot == 0 && f < F) {
        float o = acc * s;
        if (add_to) o += add_to[row * F + f];
        out[row * F + f] = o;
      }
    }
  }
}

template <int LPR>
int launch_vec(const int *rowptr, const int *col, const float *row_scale, const float *ew,
               const float *x, const float *add_to, int64_t N, int64_t F, float *out,
               const int *long_rows, const int *n_long, hipStream_t st, bool per_edge,
               const uint32_t *mask, const uint32_t *amask) {
  constexpr int G = 64 / LPR;
  const bool hubs = long_rows && n_long;
  const int long_t = hubs ? gsage_long_row_threshold() : 0;
  const int hub_blocks = hubs ? HUB_BLOCKS : 0;
  // one trip per wave: with workgroups dispatched in order the chip works on a compact window of
  // rowptr/col/out, which measured 4.5 % faster than a 16384-workgroup grid-stride loop
  const int grid = gs_grid(N, (AGG_BLOCK / 64) * G, 1 << 22) + hub_blocks;
  if (G != 2 && G != 4) mask = nullptr;   // a hint: only the 64- and 128-wide kernels act on it
  // the add_to mask is NOT a hint (masked rows of add_to may hold anything): the caller has
  // checked that this width honours it (segment_gather_impl)
#define GS_LAUNCH_VEC(WM, MK)                                                                      \
  hipLaunchKernelGGL((seg_gather_vec_k<LPR, WM, MK>), dim3(grid), dim3(AGG_BLOCK), 0, st, rowptr, \
                     col, row_scale, ew, mask, x, add_to, amask, N, F, out, long_t, long_rows,     \
                     n_long,                                                                       \
                     hub_blocks)
  if (ew && per_edge) GS_LAUNCH_VEC(2, false);   // per-entry weights: the mask is not applied
  else if (ew && mask) GS_LAUNCH_VEC(1, true);
  else if (ew) GS_LAUNCH_VEC(1, false);
  else if (mask) GS_LAUNCH_VEC(0, true);
  else GS_LAUNCH_VEC(0, false);
#undef GS_LAUNCH_VEC
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

}  // namespace

static int segment_gather_impl(const int32_t *rowptr, const int32_t *col, const float *row_scale,
                               const float *edge_weight, bool per_edge, const float *x,
                               const float *add_to, int64_t N, int64_t F, float *out,
                               const int32_t *long_rows, const int32_t *n_long,
                               const uint32_t *src_mask, gsage_stream_t stream,
                               const uint32_t *add_mask = nullptr) {
  if (N < 0 || F <= 0 || !rowptr || !x || !out) return GSAGE_ERR_INVALID_ARG;
  if (add_mask) {
    // only the masked 64- and 128-wide kernels read the add_to mask
    const bool ok = src_mask && add_to && !per_edge && (F == 64 || F == 128) &&
                    ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)add_to) & 15) == 0);
    if (!ok) return GSAGE_ERR_UNSUPPORTED;
  }
  if (N == 0) return GSAGE_OK;
  hipStream_t st = gs_stream(stream);
  bool aligned = (((uintptr_t)x | (uintptr_t)out | (uintptr_t)add_to) & 15) == 0;
  if (aligned) {
    switch (F) {
      case 4:   return launch_vec<1>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      case 8:   return launch_vec<2>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      case 16:  return launch_vec<4>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      case 32:  return launch_vec<8>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      case 64:  return launch_vec<16>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      case 128: return launch_vec<32>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      case 256: return launch_vec<64>(rowptr, col, row_scale, edge_weight, x, add_to, N, F, out, long_rows, n_long, st, per_edge, src_mask, add_mask);
      default: break;
    }
  }
  int P = 1;
  while (P < F && P < 64) P <<= 1;
  int grid = gs_grid(N, AGG_BLOCK / 64, 256 * 64);
  if (edge_weight && per_edge)
    hipLaunchKernelGGL((seg_gather_scalar_k<2>), dim3(grid), dim3(AGG_BLOCK), 0, st, rowptr, col,
                       row_scale, edge_weight, x, add_to, N, F, out, P);
  else if (edge_weight)
    hipLaunchKernelGGL((seg_gather_scalar_k<1>), dim3(grid), dim3(AGG_BLOCK), 0, st, rowptr, col,
                       row_scale, edge_weight, x, add_to, N, F, out, P);
  else
    hipLaunchKernelGGL((seg_gather_scalar_k<0>), dim3(grid), dim3(AGG_BLOCK), 0, st, rowptr, col,
                       row_scale, edge_weight, x, add_to, N, F, out, P);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_segment_gather(const int32_t *rowptr, const int32_t *col,
                                    const float *row_scale, const float *edge_weight,
                                    const float *x, const float *add_to, int64_t N, int64_t F,
                                    float *out, const int32_t *long_rows, const int32_t *n_long,
                                    gsage_stream_t stream) {
  return segment_gather_impl(rowptr, col, row_scale, edge_weight, false, x, add_to, N, F, out,
                             long_rows, n_long, nullptr, stream);
}

extern "C" int gsage_segment_gather_skip(const int32_t *rowptr, const int32_t *col,
                                         const float *row_scale, const float *edge_weight,
                                         const float *x, const float *add_to, int64_t N, int64_t F,
                                         float *out, const int32_t *long_rows,
                                         const int32_t *n_long, const uint32_t *src_nz_mask,
                                         gsage_stream_t stream) {
  return segment_gather_impl(rowptr, col, row_scale, edge_weight, false, x, add_to, N, F, out,
                             long_rows, n_long, src_nz_mask, stream);
}

extern "C" int gsage_segment_gather_live(const int32_t *rowptr, const int32_t *col,
                                         const float *row_scale, const float *edge_weight,
                                         const float *x, const float *add_to, int64_t N, int64_t F,
                                         float *out, const int32_t *long_rows,
                                         const int32_t *n_long, const uint32_t *live_mask,
                                         gsage_stream_t stream) {
  if (!live_mask || !add_to) return GSAGE_ERR_INVALID_ARG;
  return segment_gather_impl(rowptr, col, row_scale, edge_weight, false, x, add_to, N, F, out,
                             long_rows, n_long, live_mask, stream, live_mask);
}

extern "C" int gsage_segment_gather_ew(const int32_t *rowptr, const int32_t *col,
                                       const float *edge_w, const float *x, int64_t N, int64_t F,
                                       float *out, const int32_t *long_rows,
                                       const int32_t *n_long, gsage_stream_t stream) {
  if (!edge_w) return GSAGE_ERR_INVALID_ARG;
  return segment_gather_impl(rowptr, col, nullptr, edge_w, true, x, nullptr, N, F, out, long_rows,
                             n_long, nullptr, stream);
}
