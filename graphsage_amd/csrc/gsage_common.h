// Shared helpers for the gfx950 kernels of libgsage.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "gsage.h"

#define GSAGE_WAVE 64

#define GSAGE_CHECK_LAUNCH()                                   \
  do {                                                         \
    if (hipGetLastError() != hipSuccess) return GSAGE_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t gs_stream(gsage_stream_t s) { return (hipStream_t)s; }

static inline size_t gs_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Carve a workspace: returns pointer at current offset, advances by bytes (256-B aligned).
struct GsCarver {
  char *base;
  size_t off;
  explicit GsCarver(void *p) : base((char *)p), off(0) {}
  template <typename T> T *take(size_t count) {
    T *r = (T *)(base + off);
    off += gs_align_up(count * sizeof(T), 256);
    return r;
  }
};

__device__ __forceinline__ int gs_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float gs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum / max over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), the result in every lane:
// four rotations within the row (v_mov_dpp row_ror: vector instructions, no trip through the LDS
// crossbar as __shfl_xor takes).  The additions pair up exactly as the xor butterfly's do, so the
// result is bitwise the one that form gives.
template <int N> __device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); v = fmaxf(v, row_ror<1>(v));
  return v;
}

// Grid size for a grid-stride kernel: enough blocks to fill 256 CUs, capped.
static inline int gs_grid(int64_t work_items, int per_block, int cap = 256 * 8) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// Zero `words` 32-bit words with a kernel of our own.  Used instead of hipMemsetAsync wherever a
// call may be recorded into a HIP graph: a 4-byte memset node in front of the loss kernel was
// observed to leave a stale word about once in 2 500 replays of a captured training step.
static __global__ void __launch_bounds__(256) gs_zero_words_k(uint32_t *p, int64_t words) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = 0u;
}

static inline bool gs_zero_async(void *p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return true;
  const int64_t words = (int64_t)((bytes + 3) / 4);   // buffers here are all 4-byte multiples
  hipLaunchKernelGGL(gs_zero_words_k, dim3(gs_grid(words, 256 * 4)), dim3(256), 0, st, (uint32_t *)p,
                     words);
  return hipGetLastError() == hipSuccess;
}
