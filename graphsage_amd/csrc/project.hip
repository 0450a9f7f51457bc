// K3: SAGEConv projections for gfx950, forward and backward.
//
//   fwd : out = epi( [agg | x] . [W_l | W_r]^T + b_l )      epi in {none, relu, relu+dropout, l2norm}
//   bwd : dpre = epi'(grad_out);  [grad_agg | grad_x] = dpre . [W_l | W_r];
//         [grad_W_l | grad_W_r] = dpre^T . [agg | x];  grad_b = colsum(dpre)
//
// Two code paths, chosen by shape only (there is no CPU or library fallback):
//   * MFMA path (Fi, Fo in {64,128,256}): fp32-input v_mfma_f32_32x32x2_f32, 64-row
//     tiles staged through LDS in 32-deep K chunks with register prefetch; the result
//     is a k-ordered fp32 fma chain, i.e. exact fp32 like the reference's F.linear.
//   * VALU path (any shape; the 7-wide first layer uses its small-K specialisation
//     with the weights held in registers).
// Weight gradients are accumulated per workgroup into partial slabs and summed in slab
// order by a second kernel: no atomics, bitwise reproducible.
#include <atomic>

#include "gsage_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float L2_EPS = 1e-12f;

// ------------------------------------------------------------------ dropout hash RNG
// keep(seed, idx) with idx = row*Fo + col; mirrored in graphsage_amd/rng.py for tests.
// 32-bit arithmetic only (64-bit multiplies are emulated on the VALU): a bijective scramble
// of the element index keyed by the seed, then murmur3's fmix32 avalanche.
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, float p) {
  uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t h = (uint32_t)seed ^ (lo * 0x9E3779B1u);
  h ^= (uint32_t)(seed >> 32) + hi * 0x85EBCA77u;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  float u = (float)(h >> 8) * (1.0f / 16777216.0f);  // 24 random bits -> [0,1)
  return u >= p;
}

// The same decisions for four consecutive elements whose indices fit 32 bits (idx + 3 < 2^32), at about
// half the instructions: with hi = 0 the seed enters as ONE word (seed_lo ^ seed_hi), lo * C of the
// three followers is the first product plus multiples of C (v_mul_lo_u32 is a quarter-rate
// instruction), and u >= p on u = k 2^-24 is k >= ceil(p 2^24) on the integer k = h >> 8, i.e.
// h >= ceil(p 2^24) << 8 (p < 1, so the threshold fits).  Bit for bit the masks of dropout_keep.
__device__ __forceinline__ uint32_t dropout_threshold(float p) { return ((uint32_t)ceilf(p * 16777216.0f)) << 8; }
__device__ __forceinline__ void dropout_keep4_lo(uint64_t seed, uint32_t idx, uint32_t thr, bool (&k)[4]) {
  const uint32_t key = (uint32_t)seed ^ (uint32_t)(seed >> 32);
  const uint32_t m0 = idx * 0x9E3779B1u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t h = key ^ (m0 + (uint32_t)i * 0x9E3779B1u);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    k[i] = h >= thr;
  }
}

struct EpiArgs {
  int mode;
  const float *bias;
  const uint8_t *keep_mask;
  uint64_t seed;
  float p;
  float inv_keep;
  float *rnorm;
  const float *row_scale;  // dX: scale of destination 1
  unsigned *amax_out;      // nullable: max |out| as float bits (atomicMax), for the next layer's fp16 scaling
};

// largest |value| a wave stored -> *dst (float bits compare like unsigned integers; NaN above Inf).
// One L2 read per wave, an atomic only from a wave that raises the maximum.
__device__ __forceinline__ void amax_publish(unsigned m, unsigned *dst) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, m);
}

// ------------------------------------------------------------------------ weight packs
// fwd pack: Bp[n][k] (n < Fo, k < 2Fi) = k < Fi ? W_l[n][k] : W_r[n][k-Fi]
__global__ void pack_fwd_k(const float *__restrict__ wl, const float *__restrict__ wr, int Fi,
                           int Fo, float *__restrict__ bp) {
  int total = Fo * 2 * Fi;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int n = e / (2 * Fi), k = e % (2 * Fi);
    bp[e] = k < Fi ? wl[n * Fi + k] : wr[n * Fi + k - Fi];
  }
}
// dX pack: Bp[n][k] (n < 2Fi, k < Fo) = n < Fi ? W_l[k][n] : W_r[k][n-Fi]
__global__ void pack_dx_k(const float *__restrict__ wl, const float *__restrict__ wr, int Fi,
                          int Fo, float *__restrict__ bp) {
  int total = 2 * Fi * Fo;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int n = e / Fo, k = e % Fo;
    bp[e] = n < Fi ? wl[k * Fi + n] : wr[k * Fi + n - Fi];
  }
}
// VALU pack: Wt[k][j] (k < 2Fi, j < Fo) = k < Fi ? W_l[j][k] : W_r[j][k-Fi]
__global__ void pack_t_k(const float *__restrict__ wl, const float *__restrict__ wr, int Fi,
                         int Fo, float *__restrict__ wt) {
  int total = 2 * Fi * Fo;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int k = e / Fo, j = e % Fo;
    wt[e] = k < Fi ? wl[j * Fi + k] : wr[j * Fi + k - Fi];
  }
}

// =============================================================================== MFMA
struct GemmArgs {
  const float *A1, *A2;  // A = [A1 | A2], A1 is [M,K1], A2 is [M,K-K1] (A2 unused if K1 == K)
  const float *Bp;       // [Ncols, K] row-major
  float *C1, *C2;        // blockIdx.y*BN <  N1 -> C1 [M,N1];  else C2 [M,Ncols-N1]
  int64_t M;
  int K1, K, N1, Ncols;
  EpiArgs epi;
  // column-blocked operands (gemm_split_k only): element (row, col) of A1/A2 resp. C1 sits at
  // (col / 64) * blk + row * 64 + col % 64, i.e. the matrix is kept as Ncols/64 separate
  // [M, 64] tables; 0 = plain row-major
  int64_t a_blk, c_blk;
};

// staging registers use the native vector type: a float4 (struct) that is only copied in and
// out becomes a memcpy pair through a private array, i.e. scratch traffic plus a wait on the
// prefetch right behind its loads
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int G_BK = 32;
constexpr int G_PITCH = G_BK + 4;

__device__ __forceinline__ float f4c(const float4 &v, int s) {
  return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w;
}

constexpr int MODE_DX = 100;  // epilogue of the dX GEMM (scale dest 1 rows, no bias)

template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256) gemm_rows_k(GemmArgs a) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NA = BM / 32, NB = BN / 32;  // float4 per thread per chunk
  constexpr int CP = BN + 4;
  extern __shared__ __align__(16) float smem[];
  float *As = smem;
  float *Bs = smem + BM * G_PITCH;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w / WN, wn = w % WN;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = a.K, K1 = a.K1, K2 = a.K - a.K1;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NA];
  f32x4v rb[NB];
  auto gload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int f = t + i * 256;
      int r = f >> 3, c4 = f & 7;
      int64_t grow = row0 + r;
      int k = kc + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow < a.M) {
        const float *src = (k < K1) ? a.A1 + grow * K1 + k : a.A2 + grow * K2 + (k - K1);
        v = *(const float4 *)src;
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int f = t + i * 256;
      int n = f >> 3, c4 = f & 7;
      rb[i] = *(const f32x4v *)(a.Bp + (int64_t)(n0 + n) * K + kc + c4 * 4);
    }
  };
  auto sstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int f = t + i * 256;
      *(float4 *)(As + (f >> 3) * G_PITCH + (f & 7) * 4) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int f = t + i * 256;
      *(f32x4v *)(Bs + (f >> 3) * G_PITCH + (f & 7) * 4) = rb[i];
    }
  };

  gload(0);
  for (int kc = 0; kc < K; kc += G_BK) {
    sstore();
    __syncthreads();
    if (kc + G_BK < K) gload(kc + G_BK);
#pragma unroll
    for (int g = 0; g < G_BK / 8; ++g) {
      float4 af[TM], bf[TN];
      const int koff = g * 8 + (lane >> 5) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *(const float4 *)(As + ((wm * TM + i) * 32 + (lane & 31)) * G_PITCH + koff);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *(const float4 *)(Bs + ((wn * TN + j) * 32 + (lane & 31)) * G_PITCH + koff);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float av = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
            float bv = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
    }
    __syncthreads();
  }

  // accumulators -> LDS tile Cs[BM][CP]
  float *Cs = smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int col = (wn * TN + j) * 32 + (lane & 31);
        Cs[row * CP + col] = acc[i][j][r];
      }
  __syncthreads();

  // row-wise epilogue: BN/4 lanes per row, float4 each
  constexpr int LPRo = BN / 4;
  constexpr int RP = 256 / LPRo;
  const int rsub = t / LPRo, c = t % LPRo;
  const EpiArgs &e = a.epi;
  const bool second = n0 >= a.N1;
  float *C = second ? a.C2 : a.C1;
  const int ldc = second ? (a.Ncols - a.N1) : a.N1;
  const int cbase = second ? (n0 - a.N1) : n0;
  for (int r = rsub; r < BM; r += RP) {
    int64_t grow = row0 + r;
    bool live = grow < a.M;
    float4 v = *(const float4 *)(Cs + r * CP + c * 4);
    int col = cbase + c * 4;
    if (e.mode == MODE_DX) {
      if (!second && e.row_scale && live) {
        float s = e.row_scale[grow];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      }
    } else {
      float4 b = *(const float4 *)(e.bias + col);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (e.mode == GSAGE_EPI_RELU_DROPOUT && live) {
        uint64_t idx = (uint64_t)grow * (uint64_t)ldc + (uint64_t)col;
        bool k0, k1, k2, k3;
        if (e.keep_mask) {
          uchar4 m = *(const uchar4 *)(e.keep_mask + idx);
          k0 = m.x; k1 = m.y; k2 = m.z; k3 = m.w;
        } else {
          k0 = dropout_keep(e.seed, idx + 0, e.p); k1 = dropout_keep(e.seed, idx + 1, e.p);
          k2 = dropout_keep(e.seed, idx + 2, e.p); k3 = dropout_keep(e.seed, idx + 3, e.p);
        }
        v.x = k0 ? v.x * e.inv_keep : 0.f; v.y = k1 ? v.y * e.inv_keep : 0.f;
        v.z = k2 ? v.z * e.inv_keep : 0.f; v.w = k3 ? v.w * e.inv_keep : 0.f;
      }
      if (e.mode == GSAGE_EPI_L2NORM) {
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
        for (int o = 1; o < LPRo; o <<= 1) ss += __shfl_xor(ss, o, 64);
        float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
        v.x *= rn; v.y *= rn; v.z *= rn; v.w *= rn;
        if (live && c == 0 && e.rnorm) e.rnorm[grow] = rn;
      }
    }
    if (live) *(float4 *)(C + grow * ldc + col) = v;
  }
}

// hipFuncSetAttribute is per device: remember it per (kernel instantiation, device), so a process
// that drives several GPUs raises the dynamic-LDS limit on each of them.  An atomic flag: two host
// threads may both set the attribute (harmless), none may skip it.
template <typename K>
static int allow_dynamic_lds(K kernel, size_t lds, std::atomic<bool> (&done)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return GSAGE_ERR_LAUNCH;
  if (!done[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return GSAGE_ERR_LAUNCH;
    done[dev].store(true, std::memory_order_release);
  }
  return GSAGE_OK;
}

template <int WM, int WN, int TM, int TN>
int launch_gemm(const GemmArgs &a, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int stage = (BM + BN) * G_PITCH, ctile = BM * (BN + 4);
  constexpr size_t lds = sizeof(float) * (stage > ctile ? stage : ctile);
  static std::atomic<bool> done[64];
  if (int rc = allow_dynamic_lds(gemm_rows_k<WM, WN, TM, TN>, lds, done)) return rc;
  dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)(a.Ncols / BN));
  hipLaunchKernelGGL((gemm_rows_k<WM, WN, TM, TN>), grid, dim3(256), lds, st, a);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}


// Epilogue of the wide kernels, straight on the accumulators: acc[i][j][r] is row
// wm*64 + i*32 + (r&3) + 8(r>>2) + 4(lane>>5), column (wn*TN + j)*32 + (lane&31) of the tile
// (the C layout is the same for the f32 and the bf16 32x32 MFMAs).
// cs (nullable): factor per output column applied to the raw accumulators first (the fp16-split
// kernel computes on operands scaled by powers of two).
template <int TN, int NWN>   // NWN = column waves of the workgroup (they share the row norms)
__device__ __forceinline__ void wide_epilogue(f32x16 (&acc)[2][TN], const GemmArgs &a, int64_t row0,
                                              int n0, int wm, int wn, int lane, float (*rowss)[4],
                                              const float *__restrict__ cs) {
  constexpr int TM = 2;
  const EpiArgs &e = a.epi;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);   // global output column
    if (cs) {
      const float sc = cs[n];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= sc;
    }
    if (e.mode != MODE_DX) {
      const float b = e.bias[n];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + b;
          if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) v = fmaxf(v, 0.f);
          if (e.mode == GSAGE_EPI_RELU_DROPOUT) {
            int64_t grow = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (grow < a.M) {
              uint64_t idx = (uint64_t)grow * (uint64_t)a.Ncols + (uint64_t)n;
              bool keep = e.keep_mask ? (e.keep_mask[idx] != 0) : dropout_keep(e.seed, idx, e.p);
              v = keep ? v * e.inv_keep : 0.f;
            }
          }
          acc[i][j][r] = v;
        }
    }
  }
  if (e.mode == GSAGE_EPI_L2NORM) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) ss += acc[i][j][r] * acc[i][j][r];
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
        ss += __shfl_xor(ss, 8, 64); ss += __shfl_xor(ss, 16, 64);
        if ((lane & 31) == 0)
          rowss[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][wn] = ss;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float ss = rowss[lr][0];
#pragma unroll
        for (int q = 1; q < NWN; ++q) ss += rowss[lr][q];
        float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][r] *= rn;
        if (wn == 0 && (lane & 31) == 0 && e.rnorm && row0 + lr < a.M) e.rnorm[row0 + lr] = rn;
      }
  }
  unsigned amx = 0u;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    const bool second = n >= a.N1;                       // uniform over the wave (N1 % 32 == 0)
    float *C = second ? a.C2 : a.C1;
    const int ldc = second ? (a.Ncols - a.N1) : a.N1;
    const int col = second ? n - a.N1 : n;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int64_t grow = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (grow < a.M) {
          float v = acc[i][j][r];
          if (e.mode == MODE_DX && !second && e.row_scale) v *= e.row_scale[grow];
          amx = max(amx, __float_as_uint(v) & 0x7FFFFFFFu);
          if (a.c_blk && !second) C[(int64_t)(col >> 6) * a.c_blk + grow * 64 + (col & 63)] = v;
          else C[grow * ldc + col] = v;
        }
      }
  }
  if (e.amax_out) amax_publish(amx, e.amax_out);
}

// ---- wide layers (128 / 256 columns): 128-row tiles, 8 waves (2 x 4), each wave 64 rows x
// 32*TN columns.  Twice the rows per weight-tile fetch of gemm_rows_k (the 256-wide layer was
// bound by re-reading W from L2 per 64-row tile), and the epilogue works straight on the
// accumulators (row norms by lane shuffles + a 2 KiB LDS exchange between the 4 column waves),
// so no 128 x 256 output tile has to fit in LDS.
template <int TN>
__global__ void __launch_bounds__(512, 4) gemm_wide_k(GemmArgs a) {
  constexpr int BM = 128, BN = 128 * TN, TM = 2;
  constexpr int NA = BM * 8 / 512, NB = BN * 8 / 512;
  constexpr int STAGE = (BM + BN) * G_PITCH;             // floats per LDS stage (A then B)
  extern __shared__ __align__(16) float wide_smem[];     // one stage + rowss
  float (*rowss)[4] = (float (*)[4])(wide_smem + 1 * STAGE);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 2, wn = w & 3;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = a.K, K1 = a.K1, K2 = a.K - a.K1;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (macros, not lambdas: with the staging registers captured by a lambda they were kept in
  // scratch, which also put a wait on the prefetch right behind its loads)
  float4 ra[NA];
  f32x4v rb[NB];
#define WIDE_GLOAD(KC)                                                                        \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      int f = t + i * 512;                                                                    \
      int64_t grow = row0 + (f >> 3);                                                         \
      int k = (KC) + (f & 7) * 4;                                                             \
      bool in = grow < a.M;                                                                   \
      grow = in ? grow : 0;                                                                   \
      const float *src = (k < K1) ? a.A1 + grow * K1 + k : a.A2 + grow * K2 + (k - K1);       \
      ra[i] = *(const float4 *)src;                                                           \
      if (!in) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
    }                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                          \
      int f = t + i * 512;                                                                    \
      rb[i] = *(const f32x4v *)(a.Bp + (int64_t)(n0 + (f >> 3)) * K + (KC) + (f & 7) * 4);   \
    }                                                                                         \
  }
#define WIDE_SSTORE()                                                                         \
  {                                                                                           \
    float *As_ = wide_smem, *Bs_ = As_ + BM * G_PITCH;                                        \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      int f = t + i * 512;                                                                    \
      *(float4 *)(As_ + (f >> 3) * G_PITCH + (f & 7) * 4) = ra[i];                            \
    }                                                                                         \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                          \
      int f = t + i * 512;                                                                    \
      *(f32x4v *)(Bs_ + (f >> 3) * G_PITCH + (f & 7) * 4) = rb[i];                            \
    }                                                                                         \
  }

  // chunk k+1 travels global -> registers during chunk k's MFMAs.  One LDS stage and two
  // workgroups per CU measured faster (27.6 ms per 256->256 layer at 10 M rows) than two
  // stages with one barrier per chunk but a single workgroup per CU (29.0 ms).
  WIDE_GLOAD(0)
  WIDE_SSTORE()
  __syncthreads();
  for (int kc = 0; kc < K; kc += G_BK) {
    const bool more = kc + G_BK < K;
    if (more) WIDE_GLOAD(kc + G_BK)
    const float *As = wide_smem, *Bs = As + BM * G_PITCH;
#pragma unroll
    for (int g = 0; g < G_BK / 8; ++g) {
      float4 af[TM], bf[TN];
      const int koff = g * 8 + (lane >> 5) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *(const float4 *)(As + (wm * 64 + i * 32 + (lane & 31)) * G_PITCH + koff);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        bf[j] = *(const float4 *)(Bs + ((wn * TN + j) * 32 + (lane & 31)) * G_PITCH + koff);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af[i], s2), f4c(bf[j], s2),
                                                             acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) WIDE_SSTORE()
    __syncthreads();
  }
#undef WIDE_GLOAD
#undef WIDE_SSTORE

  wide_epilogue<TN, 4>(acc, a, row0, n0, wm, wn, lane, rowss, nullptr);
}

template <int TN>
int launch_gemm_wide(const GemmArgs &a, hipStream_t st) {
  constexpr int BN = 128 * TN;
  constexpr size_t lds = sizeof(float) * (1 * (128 + BN) * G_PITCH + 128 * 4);
  static std::atomic<bool> done[64];
  if (int rc = allow_dynamic_lds(gemm_wide_k<TN>, lds, done)) return rc;
  dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)(a.Ncols / BN));
  hipLaunchKernelGGL((gemm_wide_k<TN>), grid, dim3(512), lds, st, a);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

// ---- the same wide layers on the bf16 matrix pipe at fp32 accuracy.
// An fp32 number is the exact sum of three bf16 numbers (its mantissa cut into 8+8+8 bits:
// x1 = top 16 bits of x, x2 = top 16 bits of x - x1, x3 = top 16 bits of x - x1 - x2; the
// subtractions are exact).  A product a*b is then the sum of nine bf16 products, each exact in
// the MFMA's fp32 accumulator; the six with the largest weight (a1b1, a1b2, a2b1, a1b3, a2b2,
// a3b1) are kept, which leaves a relative error of ~2^-23 per product, the size of fp32's own
// rounding.  Six v_mfma_f32_32x32x16_bf16 (32 cycles for 16 k) do the work of eight
// v_mfma_f32_32x32x2_f32 (64 cycles for 2 k): 2.7x less matrix-pipe time for the same result
// to within fp32 rounding.  The weights are split once per call by split_planes_k, the
// activations as they are staged into LDS.
//   LDS holds three bf16 planes of the A tile and of the B tile in unpadded 64-byte rows (32 k),
// the four 16-byte slots of a row XOR-ed with (row >> 2) & 3, so that the ds_read_b128 fragment
// reads (lane = row, 8 consecutive k) and the stores are conflict free (SQ_LDS_BANK_CONFLICT 0).
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

// planes[p][e] = p-th bf16 part of bp[e]  (e < total)
__global__ void split_planes_k(const float *__restrict__ bp, int total, uint16_t *__restrict__ planes) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const float v = bp[e];
    const uint32_t u1 = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(u1);
    const uint32_t u2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(u2);
    planes[e] = (uint16_t)(u1 >> 16);
    planes[total + e] = (uint16_t)(u2 >> 16);
    planes[2 * total + e] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

// ---- 256-column tiles of the split scheme, built for TWO workgroups per CU: 4 waves as 2 x 2,
// each wave 64 rows x 128 columns (128 accumulator registers; 18 fragment reads per 48 MFMAs),
// one LDS stage of 74 KB.  The weight planes go global -> LDS directly (global_load_lds_dwordx4:
// no staging registers; one wave instruction writes 16 rows of 64 bytes, the slot swizzle is
// applied to the SOURCE address), the activations are fetched into registers under the MFMAs,
// split before the barrier and written after it.  Two barriers per 32-k chunk; the second
// workgroup of the CU computes while this one stages.  (Measured and dropped: 16-k steps with the
// two halves of the stage double-buffered and the DMA issued a step ahead -- one barrier per 48
// MFMAs instead of two per 96 -- 17.1 ms against 14.7 ms per 256 -> 256 layer at 10 M rows.)
template <int TN>   // 32-column blocks per wave: 4 -> 256-column tiles, 2 -> 128-column tiles
__device__ __forceinline__ void split3_tile(const GemmArgs &a, const uint16_t *__restrict__ Bpl,
                                            const int64_t row0, const int n0) {
  constexpr int BM = 128, BN = 64 * TN, TM = 2;
  constexpr int NA = BM * 8 / 256;                       // float4 of A per thread per chunk
  constexpr int PLANE = (BM + BN) * 32;                  // bf16 per plane (A rows, then B rows)
  extern __shared__ __align__(16) uint16_t s256_smem[];  // 3 planes + rowss
  float (*rowss)[4] = (float (*)[4])(s256_smem + 3 * PLANE);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int K = a.K, K1 = a.K1, K2 = a.K - a.K1;
  const int64_t plane_elems = (int64_t)a.Ncols * K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4v ra[NA];
  u32x2v pk[NA][3];
#define S256_GLOAD_A(KC)                                                                      \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      const int f = t + i * 256;                                                              \
      int64_t grow = row0 + (f >> 3);                                                         \
      const int k = (KC) + (f & 7) * 4;                                                       \
      grow = grow < a.M ? grow : 0;      /* rows past M: zeroed when split, not here */       \
      const int kk = k < K1 ? k : k - K1;                                                     \
      const float *src = a.a_blk                                                              \
          ? (k < K1 ? a.A1 : a.A2) + (int64_t)(kk >> 6) * a.a_blk + grow * 64 + (kk & 63)     \
          : (k < K1 ? a.A1 + grow * K1 + k : a.A2 + grow * K2 + (k - K1));                    \
      ra[i] = *(const f32x4v *)src;                                                           \
    }                                                                                         \
  }
#define S256_SPLIT_A()                                                                        \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      const bool in = row0 + ((t + i * 256) >> 3) < a.M;                                      \
      uint32_t p1[4], p2[4], p3[4];                                                           \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                         \
        const float v = in ? ra[i][c] : 0.f;                                                  \
        p1[c] = __float_as_uint(v) & 0xFFFF0000u;                                             \
        const float r1 = v - __uint_as_float(p1[c]);                                          \
        p2[c] = __float_as_uint(r1) & 0xFFFF0000u;                                            \
        p3[c] = __float_as_uint(r1 - __uint_as_float(p2[c])) & 0xFFFF0000u;                   \
      }                                                                                       \
      pk[i][0] = (u32x2v){(p1[0] >> 16) | p1[1], (p1[2] >> 16) | p1[3]};                      \
      pk[i][1] = (u32x2v){(p2[0] >> 16) | p2[1], (p2[2] >> 16) | p2[3]};                      \
      pk[i][2] = (u32x2v){(p3[0] >> 16) | p3[1], (p3[2] >> 16) | p3[3]};                      \
    }                                                                                         \
  }
#define S256_STORE_A()                                                                        \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      const int f = t + i * 256;                                                              \
      const int row = f >> 3, c4 = f & 7;                                                     \
      const int off = row * 32 + (((c4 >> 1) ^ ((row >> 2) & 3)) * 8) + (c4 & 1) * 4;         \
      _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                        \
        *(u32x2v *)(s256_smem + pl * PLANE + off) = pk[i][pl];                                \
    }                                                                                         \
  }
  // lane l of a wave instruction lands at LDS base + 16 l bytes = row (l >> 2), physical slot
  // (l & 3) of a 16-row block: it fetches the logical slot that belongs there
#define S256_GLDS_B(KC)                                                                       \
  {                                                                                           \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                          \
      _Pragma("unroll") for (int q = 0; q < TN; ++q) {                                        \
        const int blk = q * 4 + w;                                                            \
        const int n = blk * 16 + (lane >> 2);                                                 \
        const int sl = (lane & 3) ^ ((n >> 2) & 3);                                           \
        const uint16_t *src = Bpl + pl * plane_elems + (int64_t)(n0 + n) * K + (KC) + sl * 8; \
        uint16_t *dst = s256_smem + pl * PLANE + (BM + blk * 16) * 32;                        \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, \
                                         (__attribute__((address_space(3))) void *)dst, 16,   \
                                         0, 0);                                               \
      }                                                                                       \
  }

  S256_GLOAD_A(0)
  S256_GLDS_B(0)
  S256_SPLIT_A()
  S256_STORE_A()
  __syncthreads();
  for (int kc = 0; kc < K; kc += 32) {
    const bool more = kc + 32 < K;
    if (more) S256_GLOAD_A(kc + 32)
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {
      const int slot = ks * 2 + (lane >> 5);
      bf16x8 af[TM][3];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * 64 + i * 32 + (lane & 31);
        const int off = row * 32 + ((slot ^ ((row >> 2) & 3)) * 8);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[i][pl] = *(const bf16x8 *)(s256_smem + pl * PLANE + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = (wn * TN + j) * 32 + (lane & 31);
        const int off = (BM + n) * 32 + ((slot ^ ((n >> 2) & 3)) * 8);
        bf16x8 bf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[pl] = *(const bf16x8 *)(s256_smem + pl * PLANE + off);
        // smallest terms first, so that they are not rounded away against the big partial sums
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], c, 0, 0, 0);
          acc[i][j] = c;
        }
      }
    }
    if (more) S256_SPLIT_A()      // waits for the A loads issued before the MFMAs
    __syncthreads();              // every wave has read the stage
    if (more) {
      S256_GLDS_B(kc + 32)
      S256_STORE_A()
    }
    __syncthreads();              // drains the LDS-DMA (vmcnt) and the ds_writes
  }
#undef S256_GLOAD_A
#undef S256_SPLIT_A
#undef S256_STORE_A
#undef S256_GLDS_B
  wide_epilogue<TN, 2>(acc, a, row0, n0, wm, wn, lane, rowss, nullptr);
}

template <int TN>
__global__ void __launch_bounds__(256, 2) gemm_split_k(GemmArgs a, const uint16_t *__restrict__ Bpl) {
  split3_tile<TN>(a, Bpl, (int64_t)blockIdx.x * 128, blockIdx.y * 64 * TN);
}

// ---- the wide layers on the fp16 matrix pipe: half the matrix work of the bf16 scheme.
// With an 11-bit significand TWO parts carry 22 bits (x1 = rn16(x), x2 = rn16(x - x1); the
// subtraction is exact), and three products (a1 b1, a1 b2, a2 b1) leave a relative error of
// ~3 * 2^-22 per product: between fp32's own rounding and four times it.  What fp16 lacks is
// range, so both operands are scaled by exact powers of two first:
//   * every row n of the weight pack by 2^eb[n], its own maximum going to [2^14, 2^15);
//   * the activations by ONE factor 2^ea taken from their global maximum (a pass of its own, or
//     a bound the caller knows: `amax_hint`), undone per output column by cs[n] = 2^-(ea+eb[n]).
// fp16 subnormals are honoured by the conversion and by the MFMA (tools/probes), so an element's
// absolute error is at most max(2^-22 |x|, 2^-25) in scaled units: a row of A keeps full accuracy
// while its largest element is within 2^18 of the global maximum.  Rows below that (and
// operands outside 2^+-40, or holding Inf/NaN) are not guessed at: the workgroup that meets one
// puts its tile on a list, and the bf16 kernel above recomputes the listed tiles behind this one
// (all tiles when the list overflows or the scales are unusable; a row ABOVE fp16's range after
// scaling -- a bound that was too low -- is listed the same way).  Same swizzled 64-byte LDS rows
// as gemm_split_k, two planes instead of three; the schedule is its own (gemm_half_k below).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int H_REDO_CAP = 4096;
constexpr int H_MAX_COLS = 512;
struct HalfCtl {
  float sa;                 // 2^ea
  unsigned amax_a;          // max |A| as float bits (NaN patterns compare above Inf)
  int unsafe;               // scales unusable: every tile goes to the bf16 kernel
  int redo_count;           // tiles listed; > H_REDO_CAP: list overflowed, all tiles
  float cs[H_MAX_COLS];     // 2^-(ea + eb[n])
  int redo[H_REDO_CAP];
};

__global__ void h_ctl_init_k(HalfCtl *c) {
  c->amax_a = 0u;
  c->unsafe = 0;
  c->redo_count = 0;
}

__global__ void __launch_bounds__(256) absmax_k(const float *__restrict__ p, int64_t n4,
                                                unsigned *__restrict__ out) {
  unsigned m = 0u;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const u32x4v v = *(const u32x4v *)(p + 4 * i);
    const unsigned a = max(v[0] & 0x7FFFFFFFu, v[1] & 0x7FFFFFFFu);
    const unsigned b = max(v[2] & 0x7FFFFFFFu, v[3] & 0x7FFFFFFFu);
    m = max(m, max(a, b));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// exponent that takes a maximum with biased exponent E to [2^14, 2^15); false: out of the range
// this scheme takes on (0 counts as usable, scale 1: every product with it is 0)
__device__ __forceinline__ bool h_scale_exp(unsigned bits, int *e) {
  if (bits == 0u) { *e = 0; return true; }
  const int E = (int)((bits >> 23) & 0xFFu);
  *e = 14 - (E - 127);
  return E >= 127 - 40 && E <= 127 + 40;
}

// one workgroup per row n of the weight pack: its maximum, its two scaled planes, cs[n]
__global__ void __launch_bounds__(256) h_prep_k(const float *__restrict__ bp, int K, int total,
                                                const unsigned *__restrict__ amax_hint,
                                                HalfCtl *__restrict__ c, uint16_t *__restrict__ planes) {
  __shared__ unsigned wmax[4];
  const int n = blockIdx.x, t = threadIdx.x;
  unsigned m = 0u;
  for (int k = t; k < K; k += 256) m = max(m, __float_as_uint(bp[(int64_t)n * K + k]) & 0x7FFFFFFFu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((t & 63) == 0) wmax[t >> 6] = m;
  __syncthreads();
  m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
  int ea, eb;
  const bool ok_a = h_scale_exp(amax_hint ? *amax_hint : c->amax_a, &ea);
  const bool ok_b = h_scale_exp(m, &eb);
  const float sb = ldexpf(1.f, eb);
  for (int k = t; k < K; k += 256) {
    const float v = bp[(int64_t)n * K + k] * sb;
    const _Float16 h1 = (_Float16)v;
    const _Float16 h2 = (_Float16)(v - (float)h1);
    union { _Float16 f; uint16_t u; } u1, u2;
    u1.f = h1; u2.f = h2;
    // chunk-major, in the LDS image's own order (32-k rows, 16-byte slots XOR-ed with the row's
    // swizzle): a DMA piece of the GEMM reads 1 KiB of consecutive bytes
    const int64_t off = ((int64_t)(k >> 5) * gridDim.x + n) * 32 + ((((k & 31) >> 3) ^ ((n >> 2) & 3)) * 8) + (k & 7);
    planes[off] = u1.u;
    planes[(int64_t)total + off] = u2.u;
  }
  if (t == 0) {
    c->cs[n] = ldexpf(1.f, -(ea + eb));
    if (n == 0) c->sa = ldexpf(1.f, ea);
    if (!(ok_a && ok_b)) c->unsafe = 1;
  }
}

// Epilogue for accumulators computed TRANSPOSED (MFMA operands swapped: weights as A, activations
// as B): acc[i][j][4q + e] is output row wm*64 + i*32 + (lane & 31), column
// (wn*TN + j)*32 + 8q + 4(lane >> 5) + e of the tile, i.e. every lane holds runs of four consecutive
// columns of ONE output row and stores them as 16 bytes: 32 store instructions per thread and
// tile instead of 128 (a CU retires a store instruction every ~40 cycles whatever its width, and
// 128 of them per thread cost as much as the tile's matrix work).
template <int TN, int NWN>
__device__ __forceinline__ void wide_epilogue_t(f32x16 (&acc)[2][TN], const GemmArgs &a, int64_t row0,
                                                int n0, int wm, int wn, int lane, float (*rowss)[4],
                                                const float *__restrict__ cs) {
  constexpr int TM = 2;
  const EpiArgs &e = a.epi;
  const int h = lane >> 5;
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + (wn * TN + j) * 32 + 8 * q + 4 * h;      // first of four output columns
      const float4 sc = cs ? *(const float4 *)(cs + n) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 b = e.mode != MODE_DX ? *(const float4 *)(e.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int64_t grow = row0 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = acc[i][j][4 * q + c] * f4c(sc, c) + f4c(b, c);
          if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) v = fmaxf(v, 0.f);
          if (e.mode == GSAGE_EPI_RELU_DROPOUT && grow < a.M) {
            const uint64_t idx = (uint64_t)grow * (uint64_t)a.Ncols + (uint64_t)(n + c);
            const bool keep = e.keep_mask ? (e.keep_mask[idx] != 0) : dropout_keep(e.seed, idx, e.p);
            v = keep ? v * e.inv_keep : 0.f;
          }
          acc[i][j][4 * q + c] = v;
        }
      }
    }
  if (e.mode == GSAGE_EPI_L2NORM) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss += acc[i][j][r] * acc[i][j][r];
      ss += __shfl_xor(ss, 32, 64);
      if (h == 0) rowss[wm * 64 + i * 32 + (lane & 31)][wn] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int lr = wm * 64 + i * 32 + (lane & 31);
      float ss = rowss[lr][0];
#pragma unroll
      for (int q = 1; q < NWN; ++q) ss += rowss[lr][q];
      const float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= rn;
      if (wn == 0 && h == 0 && e.rnorm && row0 + lr < a.M) e.rnorm[row0 + lr] = rn;
    }
  }
  unsigned amx = 0u;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int64_t grow = row0 + wm * 64 + i * 32 + (lane & 31);
    if (grow < a.M) {
      const float rs = (e.mode == MODE_DX && e.row_scale) ? e.row_scale[grow] : 1.f;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nj = n0 + (wn * TN + j) * 32;
        const bool second = nj >= a.N1;                    // uniform over the wave (N1 % 32 == 0)
        float *C = second ? a.C2 : a.C1;
        const int ldc = second ? (a.Ncols - a.N1) : a.N1;
        const float f = second ? 1.f : rs;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = (second ? nj - a.N1 : nj) + 8 * q + 4 * h;
          const float4 v = make_float4(acc[i][j][4 * q] * f, acc[i][j][4 * q + 1] * f,
                                       acc[i][j][4 * q + 2] * f, acc[i][j][4 * q + 3] * f);
          float *dst = (a.c_blk && !second) ? C + (int64_t)(col >> 6) * a.c_blk + grow * 64 + (col & 63)
                                            : C + grow * ldc + col;
          *(float4 *)dst = v;
          amx = max(max(amx, __float_as_uint(v.x) & 0x7FFFFFFFu), __float_as_uint(v.y) & 0x7FFFFFFFu);
          amx = max(max(amx, __float_as_uint(v.z) & 0x7FFFFFFFu), __float_as_uint(v.w) & 0x7FFFFFFFu);
        }
      }
    }
  }
  if (e.amax_out) amax_publish(amx, e.amax_out);
}

// 512 threads = 8 waves as 4 x 2, each 64 rows x 32 TN columns; ONE persistent workgroup per CU
// walks 256-row tiles, and the K loops of all its tiles run as ONE sequence of 32-k chunks over
// two LDS stages, one barrier per chunk:
//   chunk g computes on stage g&1 while   the LDS-DMA of the weights of chunk g+1,
//                                         the split + ds_write of the activations of chunk g+1
//                                         (in registers since chunk g-1) fill stage (g+1)&1,
//                                   and   the global loads of the activations of chunk g+2 start.
// The barrier waits for the DMA but leaves those last loads in flight (vmcnt counts in issue
// order: the DMA is issued first, an empty asm keeps the compiler from moving the loads above
// it).  Chunks run on across tile boundaries, so a tile's epilogue stores drain underneath the
// next tile's first chunks.  The loop is ISSUE-bound before it is anything else (two waves per
// SIMD share one issue port with their 96 MFMAs per chunk), so everything a thread needs per
// chunk is a constant offset from a handful of values computed once: LDS addresses with
// immediate offsets, global addresses as a uniform base plus a 32-bit lane offset.
#ifdef HB_PROF
__device__ unsigned long long hb_prof[2][256][8];
#define HB_STAMP(K) { const unsigned long long now_ = __builtin_readcyclecounter(); prof_[K] += now_ - last_; last_ = now_; }
#else
#define HB_STAMP(K)
#endif
#define HB_WAIT_BARRIER(VM)                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int TN>
__global__ void __launch_bounds__(512, 1) gemm_half_k(GemmArgs a, const uint16_t *__restrict__ Bpl,
                                                      HalfCtl *__restrict__ ctl) {
  constexpr int BM = 256, BN = 64 * TN, TM = 2;
  constexpr int NA = BM * 8 / 512;                       // float4 of A per thread per chunk
  constexpr int PLANE = (BM + BN) * 32;                  // fp16 per plane (A rows, then B rows)
  constexpr int STAGE = 2 * PLANE;
  extern __shared__ __align__(16) uint16_t hb_smem[];    // 2 stages x 2 planes, then rowss
  float (*rowss)[4] = (float (*)[4])(hb_smem + 2 * STAGE);
  if (ctl->unsafe) return;                               // written before this launch: uniform
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int K = a.K, K1 = a.K1;
  // ny is 1 or 2 (the launcher's half_shape_ok admits nothing else), and for exactly those two
  // values ny - 1 is both log2(ny), the shift that takes a tile number to its row block, and
  // the mask that takes it to its column block
  const int ny = a.Ncols / BN, nysh = ny - 1, nk = K >> 5;
  const int ntiles = (int)(((a.M + BM - 1) / BM) * ny);
  const int64_t plane_elems = (int64_t)a.Ncols * K;
  const float sa = ctl->sa;
  if ((int)blockIdx.x >= ntiles) return;

  // The workgroup's chunks are numbered g = 0, 1, ...: chunk g belongs to its tile number
  // g / nk, i.e. tile blockIdx.x + (g / nk) gridDim.x, and covers k = 32 ((g + rot) % nk): the
  // workgroups start their K loops at different chunks (and a tile continues where the last one
  // stopped), so that they do not all ask L2 for the same weight lines at the same moment.
  // nk is a power of two (launcher).
  const int lognk = 31 - __builtin_clz(nk);
  const int rot = (int)((blockIdx.x * 5u) & (unsigned)(nk - 1));
  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nchunks = my_tiles << lognk;
  struct Pos { int g; };
  auto tile_of = [&](const Pos &p) { return (int)blockIdx.x + (p.g >> lognk) * (int)gridDim.x; };
  auto kc_of = [&](const Pos &p) { return ((p.g + rot) & (nk - 1)) << 5; };
  auto row0_of = [&](const Pos &p) { return (int64_t)(tile_of(p) >> nysh) * BM; };
  auto n0_of = [&](const Pos &p) { return (tile_of(p) & nysh) * BN; };

  // ---- per-thread constants
  // staging of A: thread t owns 4 consecutive k (float4 number t & 7 of the 32-k row) of rows
  // (t >> 3) + 64 i; their LDS slots differ by a constant, their global rows by a stride
  const int arow = t >> 3, ac4 = t & 7;
  const uint32_t a_stride = a.a_blk ? 256u : (uint32_t)K1 * 4u;      // bytes between rows of A
  const uint32_t smem_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)hb_smem;
  const uint32_t a_lds_b =
      smem_b + 2u * (uint32_t)(arow * 32 + (((ac4 >> 1) ^ ((arow >> 2) & 3)) * 8) + (ac4 & 1) * 4);   // bytes
  // MFMA fragments: lane = row (lane & 31) of a 32-row block, 8 k at 16-byte slot 2 ks + (lane >> 5)
  // XOR-ed with the row's swizzle, which is the same for every block: one offset, and ks = 1 is ^ 16
  const int fsw = ((lane & 31) >> 2) & 3;
  const int f_off = (lane & 31) * 32 + (((fsw & 2) | ((lane >> 5) ^ (fsw & 1))) * 8);
  const uint16_t *const fa_lds = hb_smem + wm * 64 * 32;
  const uint16_t *const fb_lds = hb_smem + (BM + wn * TN * 32) * 32;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4v ra[NA];
  float rmax[NA];          // largest |a| in this thread's slice of its NA rows of the tile being staged
  bool bad = false;        // ... found a row of a finished staging that lost bits
#pragma unroll
  for (int i = 0; i < NA; ++i) rmax[i] = 0.f;

  // rows past M are read as the tile's last row (their outputs are not stored)
#define HB_GLOAD_A(P)                                                                         \
  {                                                                                           \
    const int64_t r0 = row0_of(P);                                                            \
    const int kc = kc_of(P);                                                                  \
    const bool first = kc < K1;                                                               \
    const int kk = first ? kc : kc - K1;                                                      \
    const float *b = first ? a.A1 : a.A2;                                                     \
    const char *base = (const char *)(a.a_blk ? b + (int64_t)(kk >> 6) * a.a_blk + r0 * 64 + (kk & 63) \
                                              : b + r0 * (int64_t)(first ? K1 : K - K1) + kk); \
    const int64_t left = a.M - r0;                                                            \
    const int last = (int)(left < BM ? left : BM) - 1;                                        \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      const int row = arow + 64 * i;                                                          \
      const uint32_t off = (uint32_t)(row < last ? row : last) * a_stride + (uint32_t)ac4 * 16u; \
      ra[i] = *(const f32x4v *)(base + off);                                                  \
    }                                                                                         \
  }
  // rows of a finished tile are judged when the staging moves on to the next tile
#define HB_ROWS_DONE()                                                                        \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      float m = rmax[i];                                                                      \
      m = fmaxf(m, __shfl_xor(m, 1, 64));                                                     \
      m = fmaxf(m, __shfl_xor(m, 2, 64));                                                     \
      m = fmaxf(m, __shfl_xor(m, 4, 64));                                                     \
      /* too small to keep its bits, or (a bound that was too low) beyond what fp16 holds */  \
      bad |= (m > 0.f && m * sa < 0.125f) || !(m * sa < 65504.f);                             \
      rmax[i] = 0.f;                                                                          \
    }                                                                                         \
  }
#define HB_SPLIT_STORE_A(ST)                                                                  \
  {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                          \
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rmax[i]) : "v"(ra[i][0]), "v"(ra[i][1]));    \
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(rmax[i]) : "v"(ra[i][2]), "v"(ra[i][3]));    \
      f16x4 p1, p2;                                                                           \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                         \
        p1[c] = (_Float16)(ra[i][c] * sa);                                                    \
        p2[c] = (_Float16)__builtin_fmaf(ra[i][c], sa, -(float)p1[c]);                        \
      }                                                                                       \
      /* written by asm: behind an LDS-DMA the compiler puts vmcnt(0) in front of every LDS    \
         store of its own, which would wait for the DMA (and the loads behind it) right here */ \
      asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_lds_b + (ST) * (STAGE * 2)),       \
                   "v"(__builtin_bit_cast(uint64_t, p1)), "n"(i * 64 * 64) : "memory");       \
      asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_lds_b + (ST) * (STAGE * 2)),       \
                   "v"(__builtin_bit_cast(uint64_t, p2)), "n"(PLANE * 2 + i * 64 * 64) : "memory"); \
    }                                                                                         \
  }
  // the planes are stored chunk-major in the LDS image's own order (h_prep_k): a piece is 1 KiB
  // of consecutive bytes = 16 weight rows of 32 k
#define HB_GLDS_B(P, ST)                                                                      \
  {                                                                                           \
    const uint16_t *src0 = Bpl + ((int64_t)(kc_of(P) >> 5) * a.Ncols + n0_of(P)) * 32 + lane * 8; \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                          \
      _Pragma("unroll") for (int q = 0; q < TN / 2; ++q) {                                    \
        const int blk = q * 8 + w;                                                            \
        const uint16_t *src = src0 + pl * plane_elems + blk * 16 * 32;                        \
        const uint32_t dst = smem_b + 2u * (uint32_t)((ST) * STAGE + pl * PLANE + (BM + blk * 16) * 32); \
        /* issued by asm: the compiler answers an LDS-DMA it knows of with vmcnt(0) in front   \
           of the next LDS store and the next global load, i.e. waits for it on the spot */    \
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src),        \
                     "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");                    \
      }                                                                                       \
  }
#define HB_MFMA(ST, KS)                                                                       \
  {                                                                                           \
    const int fo = (ST) * STAGE + ((KS) ? f_off ^ 16 : f_off);                                \
    f16x8 af[TM][2];                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                            \
      _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                        \
        af[i][pl] = *(const f16x8 *)(fa_lds + fo + pl * PLANE + i * 32 * 32);                 \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                          \
      f16x8 bf[2];                                                                            \
      _Pragma("unroll") for (int pl = 0; pl < 2; ++pl)                                        \
        bf[pl] = *(const f16x8 *)(fb_lds + fo + pl * PLANE + j * 32 * 32);                    \
      /* small terms first, so that they are not rounded away against the big partial sums */ \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0], af[0][1], acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0], af[1][1], acc[1][j], 0, 0, 0); \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[1], af[0][0], acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[1], af[1][0], acc[1][j], 0, 0, 0); \
      acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0], af[0][0], acc[0][j], 0, 0, 0); \
      acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[0], af[1][0], acc[1][j], 0, 0, 0); \
    }                                                                                         \
  }

  // Every workgroup has the same work per tile, and a tile ends in a burst of 256 KiB of
  // stores: started together, all CUs would write at the same moments (the stores then drain at
  // the chip's write rate while nothing computes) and read at the same moments.  Sixteen
  // start phases a sixteenth of a tile apart spread both streams evenly over time.
  for (int i = (int)(blockIdx.x & 15u) * nk; i > 0; --i) __builtin_amdgcn_s_sleep(5);
  // prologue: chunk 0 into stage 0, chunk 1's activations into registers
  {
    const Pos p0{0}, p1{1};
    HB_GLDS_B(p0, 0)
    HB_GLOAD_A(p0)
    HB_SPLIT_STORE_A(0)
    if (nk == 1) HB_ROWS_DONE()                   // a one-chunk K: chunk 1 is the next tile already
    if (1 < nchunks) HB_GLOAD_A(p1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HB_WAIT_BARRIER(0);
  }
#ifdef HB_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  for (int g = 0; g < nchunks; ++g) {
    HB_STAMP(0)
    // stage g & 1 holds chunk g; registers ra hold the activations of chunk g + 1
    const int st = g & 1;
    const Pos pc{g}, pd{g + 1}, pl2{g + 2};
    const bool have_d = g + 1 < nchunks, have_l = g + 2 < nchunks;
    // The two waves of a SIMD (w and w + 4) take the chunk's two halves in opposite order: one
    // stages while the other one's MFMAs have the matrix pipe, then they swap.  In the same
    // order both would stage together with the pipe idle, then queue for it.
    if (w < 4) {
      if (have_d) {
        HB_SPLIT_STORE_A(st ^ 1)
        if (((g + 2) & (nk - 1)) == 0) HB_ROWS_DONE() // that was the last chunk of its tile
      }
      HB_STAMP(1)
      if (have_l) HB_GLOAD_A(pl2)
      asm volatile("" ::: "memory");
      if (have_d) HB_GLDS_B(pd, st ^ 1)           // last in issue order: the compiler's own waits
                                                  // in this chunk never cover it
      HB_STAMP(2)
      HB_MFMA(st, 0)
      HB_MFMA(st, 1)
      HB_STAMP(3)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      HB_STAMP(4)
    } else {
      if (have_d) HB_GLDS_B(pd, st ^ 1)           // a whole chunk to land
      HB_STAMP(2)
      HB_MFMA(st, 0)
      HB_MFMA(st, 1)
      HB_STAMP(3)
      if (have_d) {
        HB_SPLIT_STORE_A(st ^ 1)                  // the compiler waits for ra here (vmcnt(0): the
                                                  // DMA, a chunk old, is covered too)
        if (((g + 2) & (nk - 1)) == 0) HB_ROWS_DONE()
      }
      HB_STAMP(1)
      if (have_l) {
        HB_GLOAD_A(pl2)                           // needed at the END of the next chunk
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      HB_STAMP(4)
    }
    HB_WAIT_BARRIER(0x3f);
    HB_STAMP(5)
    if (((g + 1) & (nk - 1)) == 0) {              // the tile is complete
      // `bad` as of the staging of this tile's last chunk (one chunk ago, or just now for K = 32)
      if (__syncthreads_or(bad) && t == 0) {
        // two 128-row tiles of the list kernel's tiling
        const int64_t tx = tile_of(pc) >> nysh, ty = tile_of(pc) & nysh;
        const bool two = (2 * tx + 1) * 128 < a.M;
        const int slot = atomicAdd(&ctl->redo_count, two ? 2 : 1);
        if (slot < H_REDO_CAP) ctl->redo[slot] = (int)(2 * tx * ny + ty);
        if (two && slot + 1 < H_REDO_CAP) ctl->redo[slot + 1] = (int)((2 * tx + 1) * ny + ty);
      }
      bad = false;
      wide_epilogue_t<TN, 2>(acc, a, row0_of(pc), n0_of(pc), wm, wn, lane, rowss, ctl->cs);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      HB_STAMP(6)
    }
  }
#ifdef HB_PROF
  if ((t == 0 || t == 256) && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) hb_prof[t >> 8][blockIdx.x][k] = k == 7 ? (unsigned long long)nchunks : prof_[k];
#endif
#undef HB_GLOAD_A
#undef HB_ROWS_DONE
#undef HB_SPLIT_STORE_A
#undef HB_GLDS_B
#undef HB_MFMA
}

// the tiles the fp16 kernel would not vouch for, recomputed by the bf16 scheme
template <int TN>
__global__ void __launch_bounds__(256, 2) gemm_split_list_k(GemmArgs a, const uint16_t *__restrict__ Bpl,
                                                            const HalfCtl *__restrict__ ctl) {
  const int ny = a.Ncols / (64 * TN);
  const int64_t ntiles = ((a.M + 127) / 128) * ny;
  const bool all = ctl->unsafe || ctl->redo_count > H_REDO_CAP;
  const int64_t cnt = all ? ntiles : (int64_t)ctl->redo_count;
  for (int64_t i = blockIdx.x; i < cnt; i += gridDim.x) {
    const int64_t tile = all ? i : (int64_t)ctl->redo[i];
    split3_tile<TN>(a, Bpl, (tile / ny) * 128, (int)(tile % ny) * 64 * TN);
    __syncthreads();            // the next tile's staging overwrites this one's LDS
  }
}

template <int TN>
int launch_gemm_split(const GemmArgs &a, uint16_t *planes, hipStream_t st) {
  constexpr int BN = 64 * TN;
  constexpr size_t lds = 3 * (size_t)(128 + BN) * 32 * sizeof(uint16_t) + 128 * 4 * sizeof(float);
  static std::atomic<bool> done[64];
  if (int rc = allow_dynamic_lds(gemm_split_k<TN>, lds, done)) return rc;
  const int total = a.Ncols * a.K;
  hipLaunchKernelGGL(split_planes_k, dim3(gs_grid(total, 256, 256)), dim3(256), 0, st, a.Bp, total,
                     planes);
  GSAGE_CHECK_LAUNCH();
  dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)(a.Ncols / BN));
  hipLaunchKernelGGL((gemm_split_k<TN>), grid, dim3(256), lds, st, a, (const uint16_t *)planes);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

// shapes gemm_half_k<TN> takes (TN = 2: 128-column tiles, 4: 256-column tiles)
static bool half_shape_ok(const GemmArgs &a, int BN) {
  const int nk = a.K >> 5;
  return a.Ncols <= H_MAX_COLS && a.K % 32 == 0 && a.K1 % 32 == 0 && (a.Ncols == BN || a.Ncols == 2 * BN) &&
         (a.K == a.K1 || a.K == 2 * a.K1) && nk > 0 && (nk & (nk - 1)) == 0;
}
// fp16 scheme: control block and planes live behind the fp32 pack (see pack_bytes)
template <int TN>
int launch_gemm_half(const GemmArgs &a, uint16_t *planes, HalfCtl *ctl, const unsigned *amax_hint,
                     hipStream_t st) {
  constexpr int BN = 64 * TN;
  constexpr size_t lds = 2 * 2 * (size_t)(256 + BN) * 32 * sizeof(uint16_t) + 256 * 4 * sizeof(float);
  constexpr size_t lds3 = 3 * (size_t)(128 + BN) * 32 * sizeof(uint16_t) + 128 * 4 * sizeof(float);
  if (!half_shape_ok(a, BN)) return GSAGE_ERR_UNSUPPORTED;
  static std::atomic<bool> done[64], done3[64];
  if (int rc = allow_dynamic_lds(gemm_half_k<TN>, lds, done)) return rc;
  if (int rc = allow_dynamic_lds(gemm_split_list_k<TN>, lds3, done3)) return rc;
  const int total = a.Ncols * a.K;
  hipLaunchKernelGGL(h_ctl_init_k, dim3(1), dim3(1), 0, st, ctl);
  GSAGE_CHECK_LAUNCH();
  if (!amax_hint) {
    const int64_t n1 = a.M * a.K1 / 4, n2 = a.M * (a.K - a.K1) / 4;
    hipLaunchKernelGGL(absmax_k, dim3(gs_grid(n1, 256 * 8, 2048)), dim3(256), 0, st, a.A1, n1,
                       &ctl->amax_a);
    if (n2)
      hipLaunchKernelGGL(absmax_k, dim3(gs_grid(n2, 256 * 8, 2048)), dim3(256), 0, st, a.A2, n2,
                         &ctl->amax_a);
    GSAGE_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(h_prep_k, dim3(a.Ncols), dim3(256), 0, st, a.Bp, a.K, total, amax_hint, ctl,
                     planes);
  GSAGE_CHECK_LAUNCH();
  const int64_t ntiles = ((a.M + 255) / 256) * (a.Ncols / BN);
  hipLaunchKernelGGL((gemm_half_k<TN>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(512), lds,
                     st, a, (const uint16_t *)planes, ctl);
  GSAGE_CHECK_LAUNCH();
  // the tiles it listed (normally none: the kernel below then finds an empty list and leaves)
  hipLaunchKernelGGL(split_planes_k, dim3(gs_grid(total, 256, 256)), dim3(256), 0, st, a.Bp, total,
                     planes);
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL((gemm_split_list_k<TN>), dim3(512), dim3(256), lds3, st, a,
                     (const uint16_t *)planes, (const HalfCtl *)ctl);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

int launch_gemm_bn(int BN, const GemmArgs &a, hipStream_t st) {
  switch (BN) {
    case 64:  return launch_gemm<2, 2, 1, 1>(a, st);
    case 128: return launch_gemm<1, 4, 2, 1>(a, st);
    case 256: return launch_gemm<1, 4, 2, 2>(a, st);
    default:  return GSAGE_ERR_UNSUPPORTED;
  }
}

// ---- dW = dpre^T . [agg | x] with the row index as the MFMA K dimension.
// Output tile 64 (Fo) x 128 (2Fi cols) per workgroup; grid.x = row chunks (slabs),
// grid.y = output tiles.  slab layout: [Fo][2Fi] then [Fo] bias sums.
constexpr int DW_RK = 32;  // rows per chunk
__global__ void __launch_bounds__(256)
dw_mfma_k(const float *__restrict__ dpre, const float *__restrict__ agg,
          const float *__restrict__ x, int64_t M, int Fi, int Fo, int64_t rows_per_block,
          float *__restrict__ slabs) {
  __shared__ __align__(16) float Ds[DW_RK][64];
  __shared__ __align__(16) float Ac[DW_RK][128];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int K2 = 2 * Fi;
  const int ntn = K2 / 128;
  const int mt = blockIdx.y / ntn, nt = blockIdx.y % ntn;
  const int m0 = mt * 64, c0 = nt * 128;
  int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float bsum = 0.f;

  float4 rd[2], rc[4];
  auto gload = [&](int64_t rb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int f = t + i * 256;           // 512 float4: row = f>>4, c4 = f&15
      int64_t row = rb + (f >> 4);
      rd[i] = row < r_end ? *(const float4 *)(dpre + row * Fo + m0 + (f & 15) * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = t + i * 256;           // 1024 float4: row = f>>5, c4 = f&31
      int64_t row = rb + (f >> 5);
      int n = c0 + (f & 31) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < r_end) v = n < Fi ? *(const float4 *)(agg + row * Fi + n)
                                  : *(const float4 *)(x + row * Fi + (n - Fi));
      rc[i] = v;
    }
  };
  if (r_begin < r_end) gload(r_begin);
  for (int64_t rb = r_begin; rb < r_end; rb += DW_RK) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int f = t + i * 256;
      *(float4 *)(&Ds[f >> 4][(f & 15) * 4]) = rd[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int f = t + i * 256;
      *(float4 *)(&Ac[f >> 5][(f & 31) * 4]) = rc[i];
    }
    __syncthreads();
    if (rb + DW_RK < r_end) gload(rb + DW_RK);
#pragma unroll
    for (int kk = 0; kk < DW_RK / 2; ++kk) {
      int kr = 2 * kk + (lane >> 5);
      float a0 = Ds[kr][lane & 31];
      float a1 = Ds[kr][32 + (lane & 31)];
      float b = Ac[kr][w * 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
    }
    if (nt == 0 && t < 64) {
#pragma unroll 8
      for (int k = 0; k < DW_RK; ++k) bsum += Ds[k][t];
    }
    __syncthreads();
  }
  float *slab = slabs + (int64_t)blockIdx.x * ((int64_t)Fo * K2 + Fo);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int mrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int ncol = c0 + w * 32 + (lane & 31);
    slab[(int64_t)(m0 + mrow) * K2 + ncol] = acc0[r];
    slab[(int64_t)(m0 + 32 + mrow) * K2 + ncol] = acc1[r];
  }
  if (nt == 0 && t < 64) slab[(int64_t)Fo * K2 + m0 + t] = bsum;
}


// ======================================================================= 64 -> 64 layer
// The headline shape (hidden = out = 64).  Persistent workgroups walk 64-row tiles; the
// packed weights live in REGISTERS in MFMA B-operand order for the whole launch (a wave's
// 32 output columns x K), so per tile only the activations move: global -> LDS (whole K at
// once, pitch 132 floats => conflict-free ds_read_b128) -> MFMA.  Several workgroups per CU
// overlap one tile's loads with another's MFMAs.
constexpr int S64_AP = 132;   // pitch of the [agg | x] tile (128 columns)
constexpr int S64_DP = 68;    // pitch of a 64-column tile


__global__ void __launch_bounds__(256, 2)
proj64_fwd_k(const float *__restrict__ agg, const float *__restrict__ x,
             const float *__restrict__ bp /* [64][128] */, int64_t M, EpiArgs e,
             float *__restrict__ out) {
  __shared__ __align__(16) float As[64 * S64_AP];
  __shared__ __align__(16) float Cs[64 * S64_DP];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  float4 breg[16];
#pragma unroll
  for (int g = 0; g < 16; ++g)
    breg[g] = *(const float4 *)(bp + (wn * 32 + (lane & 31)) * 128 + g * 8 + (lane >> 5) * 4);
  const int rsub = t >> 4, c4 = (t & 15) * 4;
  const float4 bias = *(const float4 *)(e.bias + c4);
  const int64_t ntiles = (M + 63) / 64;
  // operands of tile i+1 are fetched into registers underneath tile i's MFMAs and epilogue
  float4 r1[4], r2[4];
#define S64_FETCH(ROW0)                                                     \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                           \
    int64_t row = (ROW0) + p * 16 + rsub;                                   \
    bool in = row < M;                                                      \
    row = in ? row : 0;                                                     \
    r1[p] = *(const float4 *)(agg + row * 64 + c4);                         \
    r2[p] = *(const float4 *)(x + row * 64 + c4);                           \
    if (!in) { r1[p] = make_float4(0.f, 0.f, 0.f, 0.f); r2[p] = r1[p]; }    \
  }
  if ((int64_t)blockIdx.x < ntiles) { S64_FETCH((int64_t)blockIdx.x * 64) }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * 64;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      *(float4 *)(As + (p * 16 + rsub) * S64_AP + c4) = r1[p];
      *(float4 *)(As + (p * 16 + rsub) * S64_AP + 64 + c4) = r2[p];
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) { S64_FETCH((tile + gridDim.x) * 64) }
#undef S64_FETCH
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *arow = As + (wm * 32 + (lane & 31)) * S64_AP + (lane >> 5) * 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float4 af = *(const float4 *)(arow + g * 8);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af, s2), f4c(breg[g], s2), acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      Cs[row * S64_DP + wn * 32 + (lane & 31)] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      int r = pass * 16 + rsub;
      int64_t grow = row0 + r;
      bool live = grow < M;
      float4 v = *(const float4 *)(Cs + r * S64_DP + c4);
      v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
      if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (e.mode == GSAGE_EPI_RELU_DROPOUT && live) {
        uint64_t idx = (uint64_t)grow * 64u + (uint64_t)c4;
        bool k0, k1, k2, k3;
        if (e.keep_mask) {
          uchar4 m = *(const uchar4 *)(e.keep_mask + idx);
          k0 = m.x; k1 = m.y; k2 = m.z; k3 = m.w;
        } else {
          k0 = dropout_keep(e.seed, idx + 0, e.p); k1 = dropout_keep(e.seed, idx + 1, e.p);
          k2 = dropout_keep(e.seed, idx + 2, e.p); k3 = dropout_keep(e.seed, idx + 3, e.p);
        }
        v.x = k0 ? v.x * e.inv_keep : 0.f; v.y = k1 ? v.y * e.inv_keep : 0.f;
        v.z = k2 ? v.z * e.inv_keep : 0.f; v.w = k3 ? v.w * e.inv_keep : 0.f;
      }
      if (e.mode == GSAGE_EPI_L2NORM) {
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        ss = row16_sum(ss);
        float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
        v.x *= rn; v.y *= rn; v.z *= rn; v.w *= rn;
        if (live && (t & 15) == 0 && e.rnorm) e.rnorm[grow] = rn;
      }
      if (live) *(float4 *)(out + grow * 64 + c4) = v;
    }
    // the next tile's staging overwrites As only: every wave has passed the barrier above,
    // i.e. finished its MFMA reads of As; Cs is rewritten only after the next barrier
  }
}


// ---- the same 64 -> 64 forward on the fp16 matrix pipe (round 3).  proj64_fwd_k above is bound by
// its matrix work (256 fp32 MFMAs of 64 cycles per tile, pipe busy 62 % of the kernel's cycles at
// a clock the kernels around it pull down); here every fp32 operand is the sum of two fp16 parts
// (hi = rtz16(v s), lo = rtz16(v s - hi), s a power of two that takes the largest magnitude to
// 2^13..2^14), three of the four products are issued, smallest first, and a tile costs 96 MFMAs of
// 32 cycles.  The activations' scale comes from a bound handed in (amax_in: max |x| as float
// bits, published by the kernel that produced x; the mean aggregation of x cannot exceed it), the
// weights' from their own maximum (p64h_pack_k).  Absolute operand error <= max(2^-21 |v|,
// 2^-38 max|v|); the bias is added in fp32 in the epilogue.
typedef short p64h_b8 __attribute__((ext_vector_type(8)));
typedef _Float16 p64h_f8 __attribute__((ext_vector_type(8)));
typedef _Float16 p64h_f2 __attribute__((ext_vector_type(2)));
constexpr int P64H_PITCH = 136;          // halves per LDS row of a plane: 128 k + 8 (272 bytes: conflict-free b128 reads)
constexpr float P64H_TOP = 8192.f;       // 2^13

__device__ __forceinline__ float p64h_pow2_floor(float v, float fallback) {
  const uint32_t e = (__float_as_uint(v) >> 23) & 255u;
  if (!(v > 0.f) || e < 27u || e > 227u) return fallback;
  return __uint_as_float(e << 23);
}

// planes[pl][n][k] (n < 64, k < 128: W_l | W_r) as fp16 hi / lo of W s_w; params = {s_a, 1 / (s_a s_w)}
__global__ void __launch_bounds__(256) p64h_pack_k(const float *__restrict__ wl, const float *__restrict__ wr,
                                                   const uint32_t *__restrict__ amax_in,
                                                   uint16_t *__restrict__ planes, float *__restrict__ params) {
  __shared__ float red[4];
  auto value = [&](int e) { const int n = e >> 7, k = e & 127; return k < 64 ? wl[n * 64 + k] : wr[n * 64 + k - 64]; };
  float m = 0.f;
  for (int e = threadIdx.x; e < 64 * 128; e += 256) m = fmaxf(m, fabsf(value(e)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sw = P64H_TOP / p64h_pow2_floor(m, P64H_TOP);
  for (int e = threadIdx.x; e < 64 * 128; e += 256) {
    const float v = value(e) * sw;
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    planes[e] = __builtin_bit_cast(uint16_t, hi);
    planes[64 * 128 + e] = __builtin_bit_cast(uint16_t, lo);
  }
  if (threadIdx.x == 0) {
    const float sa = P64H_TOP / p64h_pow2_floor(__uint_as_float(*amax_in), 1.f);
    params[0] = sa;
    params[1] = 1.f / (sa * sw);
  }
}

__global__ void __launch_bounds__(256, 2)
proj64h_fwd_k(const float *__restrict__ agg, const float *__restrict__ x,
              const uint16_t *__restrict__ planes, const float *__restrict__ params, int64_t M,
              EpiArgs e, float *__restrict__ out) {
  __shared__ __align__(16) uint16_t Ah[2][64 * P64H_PITCH];   // hi / lo planes of the [agg | x] tile
  __shared__ __align__(16) float Cs[64 * S64_DP];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const float sa = params[0], unscale = params[1];
  // B operand of k-step s (16 k): column wn 32 + (lane & 31), k = 16 s + 8 (lane >> 5) .. + 7
  p64h_f8 bh[8], bl[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int off = (wn * 32 + (lane & 31)) * 128 + 16 * s + 8 * (lane >> 5);
    bh[s] = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(planes + off));
    bl[s] = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(planes + 64 * 128 + off));
  }
  const int rsub = t >> 4, c4 = (t & 15) * 4;
  const float4 bias = *(const float4 *)(e.bias + c4);
  const int64_t ntiles = (M + 63) / 64;
  unsigned amx = 0u;
  float4 r1[4], r2[4];
#define P64H_FETCH(ROW0)                                                    \
  _Pragma("unroll") for (int p = 0; p < 4; ++p) {                           \
    int64_t row = (ROW0) + p * 16 + rsub;                                   \
    bool in = row < M;                                                      \
    row = in ? row : 0;                                                     \
    r1[p] = *(const float4 *)(agg + row * 64 + c4);                         \
    r2[p] = *(const float4 *)(x + row * 64 + c4);                           \
    if (!in) { r1[p] = make_float4(0.f, 0.f, 0.f, 0.f); r2[p] = r1[p]; }    \
  }
  // four fp32 values -> their scaled hi and lo halves, 8 bytes each, into the two planes
#define P64H_SPLIT_STORE(V, ROW, COL)                                                              \
  {                                                                                                \
    const float v0 = (V).x * sa, v1 = (V).y * sa, v2 = (V).z * sa, v3 = (V).w * sa;                \
    const p64h_f2 h01 = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v0, v1));           \
    const p64h_f2 h23 = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v2, v3));           \
    const p64h_f2 l01 = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v0 - (float)h01[0], v1 - (float)h01[1])); \
    const p64h_f2 l23 = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v2 - (float)h23[0], v3 - (float)h23[1])); \
    uint2 hq, lq;                                                                                  \
    hq.x = __builtin_bit_cast(uint32_t, h01); hq.y = __builtin_bit_cast(uint32_t, h23);            \
    lq.x = __builtin_bit_cast(uint32_t, l01); lq.y = __builtin_bit_cast(uint32_t, l23);            \
    *(uint2 *)(Ah[0] + (ROW) * P64H_PITCH + (COL)) = hq;                                           \
    *(uint2 *)(Ah[1] + (ROW) * P64H_PITCH + (COL)) = lq;                                           \
  }
  if ((int64_t)blockIdx.x < ntiles) { P64H_FETCH((int64_t)blockIdx.x * 64) }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * 64;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      P64H_SPLIT_STORE(r1[p], p * 16 + rsub, c4)
      P64H_SPLIT_STORE(r2[p], p * 16 + rsub, 64 + c4)
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) { P64H_FETCH((tile + gridDim.x) * 64) }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int aoff = (wm * 32 + (lane & 31)) * P64H_PITCH + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const p64h_f8 ah = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(Ah[0] + aoff + 16 * s));
      const p64h_f8 al = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(Ah[1] + aoff + 16 * s));
      // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      Cs[row * S64_DP + wn * 32 + (lane & 31)] = acc[r] * unscale;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      int r = pass * 16 + rsub;
      int64_t grow = row0 + r;
      bool live = grow < M;
      float4 v = *(const float4 *)(Cs + r * S64_DP + c4);
      v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
      if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (e.mode == GSAGE_EPI_RELU_DROPOUT && live) {
        uint64_t idx = (uint64_t)grow * 64u + (uint64_t)c4;
        bool k0, k1, k2, k3;
        if (e.keep_mask) {
          uchar4 m = *(const uchar4 *)(e.keep_mask + idx);
          k0 = m.x; k1 = m.y; k2 = m.z; k3 = m.w;
        } else {
          k0 = dropout_keep(e.seed, idx + 0, e.p); k1 = dropout_keep(e.seed, idx + 1, e.p);
          k2 = dropout_keep(e.seed, idx + 2, e.p); k3 = dropout_keep(e.seed, idx + 3, e.p);
        }
        v.x = k0 ? v.x * e.inv_keep : 0.f; v.y = k1 ? v.y * e.inv_keep : 0.f;
        v.z = k2 ? v.z * e.inv_keep : 0.f; v.w = k3 ? v.w * e.inv_keep : 0.f;
      }
      if (e.mode == GSAGE_EPI_L2NORM) {
        float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        ss = row16_sum(ss);
        float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
        v.x *= rn; v.y *= rn; v.z *= rn; v.w *= rn;
        if (live && (t & 15) == 0 && e.rnorm) e.rnorm[grow] = rn;
      }
      if (live) {
        *(float4 *)(out + grow * 64 + c4) = v;
        amx = max(max(amx, __float_as_uint(v.x) & 0x7FFFFFFFu), __float_as_uint(v.y) & 0x7FFFFFFFu);
        amx = max(max(amx, __float_as_uint(v.z) & 0x7FFFFFFFu), __float_as_uint(v.w) & 0x7FFFFFFFu);
      }
    }
    // the next tile's staging overwrites Ah only: every wave has passed the barrier above, i.e.
    // finished its MFMA reads of Ah; Cs is rewritten only after the next barrier
  }
#undef P64H_FETCH
#undef P64H_SPLIT_STORE
  if (e.amax_out) amax_publish(amx, e.amax_out);
}

// Fused backward of a 64 -> 64 layer: per 64-row tile
//   dpre = epi'(grad_out)                      (registers -> LDS, never written to HBM)
//   [grad_agg | grad_x] = dpre . [W_l | W_r]   (MFMA, weights in registers; grad_agg * scale)
//   dW += dpre^T . [agg | x], db += colsum     (MFMA over the row index, kept in registers
//                                               across the workgroup's tiles -> one slab)
//
// ROWS: the tiles are cut from a LIST of row numbers (`rows`, ascending, `*n_rows` entries: the
// rows a pair batch touched) instead of from all M rows: every other row has grad_out = 0, hence
// dpre = 0, contributes nothing to dW/db, and its rows of grad_agg / grad_x -- which would be
// zero -- are NOT written (the transposed gather that follows masks them by the same bits).
//
// HDX (round 3): the dX half on the fp16 matrix pipe.  The kernel is bound by its fp32 MFMAs (64 per
// tile and wave), and dX = dpre . [W_l | W_r] takes rows one by one: every dpre row is scaled by its
// own power of two (its largest magnitude to 2^13, found while the row is staged: it sits in 16
// lanes), split into two fp16 parts as it is read back from LDS, multiplied with the weights' two
// fp16 planes (bwd64h_pack_k: scaled by their own maximum) in three v_mfma_f32_32x32x16_f16 per 16
// columns of dpre -- 12 instead of 32 fp32 MFMAs -- and unscaled row by row at the store.  dW sums
// over rows (one scale per row would have to go into the other operand), so it stays on fp32 MFMAs.
// planes[pl][n][k] (n < 128: W_l | W_r columns, k < 64) = fp16 hi / lo of W s_w; params[0] = 1 / s_w
__global__ void __launch_bounds__(256) bwd64h_pack_k(const float *__restrict__ wl, const float *__restrict__ wr,
                                                     uint16_t *__restrict__ planes, float *__restrict__ params) {
  __shared__ float red[4];
  auto value = [&](int e) { const int n = e >> 6, k = e & 63; return n < 64 ? wl[k * 64 + n] : wr[k * 64 + n - 64]; };
  float m = 0.f;
  for (int e = threadIdx.x; e < 128 * 64; e += 256) m = fmaxf(m, fabsf(value(e)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sw = P64H_TOP / p64h_pow2_floor(m, P64H_TOP);
  for (int e = threadIdx.x; e < 128 * 64; e += 256) {
    const float v = value(e) * sw;
    const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
    planes[e] = __builtin_bit_cast(uint16_t, hi);
    planes[128 * 64 + e] = __builtin_bit_cast(uint16_t, lo);
  }
  if (threadIdx.x == 0) params[0] = 1.f / sw;
}

template <int MODE, bool ROWS, bool HDX = false>   // 0 = no epilogue, 1 = ReLU(+dropout), 2 = L2 normalisation
__global__ void __launch_bounds__(256, 3)
bwd64_fused_k(const float *__restrict__ g, const float *__restrict__ y,
              const float *__restrict__ rnorm, const float *__restrict__ agg,
              const float *__restrict__ x, const float *__restrict__ bp /* [128][64] */,
              const float *__restrict__ agg_scale, int64_t M_all, float inv_keep,
              float *__restrict__ grad_agg, float *__restrict__ grad_x,
              float *__restrict__ slabs, uint32_t *__restrict__ nz_mask,
              const int *__restrict__ rows, const int *__restrict__ n_rows, int64_t rows_cap,
              unsigned *__restrict__ gmax /* nullable: max |grad_agg|, |grad_x| stored (float bits) */,
              const uint16_t *__restrict__ hplanes = nullptr, const float *__restrict__ hparams = nullptr) {
  // 32-row tiles, both LDS tiles double-buffered: one barrier per tile, 51 KB per workgroup and
  // <= 168 VGPRs, so three workgroups share a CU and their MFMA phases interleave
  constexpr int TR = 32;
  __shared__ __align__(16) float Ac[2][TR * S64_AP];
  __shared__ __align__(16) float Dp[2][TR * S64_DP];
  __shared__ float sc[2][TR];
  __shared__ int rid[2][TR];     // ROWS: the tile's row numbers, for the dX stores
  __shared__ float rsA[2][HDX ? TR : 1], rsI[2][HDX ? TR : 1];   // HDX: a dpre row's scale, and what undoes it
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int64_t M = M_all;             // ROWS: entries of the list; else rows of the operands
  if (ROWS) {
    M = *n_rows;
    if (M > rows_cap) M = rows_cap;
  }
  // dX: wave w owns output columns 32w..32w+31 of [grad_agg | grad_x] for all 32 rows
  float4 breg[HDX ? 1 : 8];
  p64h_f8 bh[HDX ? 4 : 1], bl[HDX ? 4 : 1];
  float inv_sw = 1.f;
  if (HDX) {
    // B operand of k-step s: column 32 w + (lane & 31), k = 16 s + 8 (lane >> 5) .. + 7 of both planes
    inv_sw = hparams[0];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int off = (w * 32 + (lane & 31)) * 64 + 16 * s2 + 8 * (lane >> 5);
      bh[s2] = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(hplanes + off));
      bl[s2] = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(hplanes + 128 * 64 + off));
    }
  } else {
#pragma unroll
    for (int gi = 0; gi < 8; ++gi)
      breg[gi] = *(const float4 *)(bp + (w * 32 + (lane & 31)) * 64 + gi * 8 + (lane >> 5) * 4);
  }
  f32x16 dw0, dw1;   // dW tiles: rows m = 0..31 / 32..63, columns 32w .. 32w+31 of [agg | x]
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw0[r] = 0.f; dw1[r] = 0.f; }
  float bsum = 0.f;
  // (the running max |dX| lives in LDS: the kernel sits at its register limit)
  __shared__ unsigned gmx[256];
  gmx[t] = 0u;
  const int rsub = t >> 4, c4 = (t & 15) * 4;
  const int64_t ntiles = (M + TR - 1) / TR;
  // the tile's operands travel through registers: the loads of tile i+1 are issued right after
  // tile i's operands reach LDS and complete underneath tile i's MFMAs
  float4 gv[2], yv[2], av[2], xv[2];
  float rn[2], scv = 1.f;
  // ROWS: the row numbers of the tile after the one in flight are fetched a tile ahead (nid),
  // those of the tile being fetched sit in cid and go to LDS with its operands
  int nid[2] = {0, 0}, cid[2] = {0, 0}, srow = 0, nsrow = 0;
#define BWD64_IDS(ROW0)                                                            \
  if (ROWS) {                                                                      \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                \
      const int64_t pos = (ROW0) + rsub * 2 + p;                                   \
      nid[p] = rows[pos < M ? pos : 0];                                            \
    }                                                                              \
    if (t < TR) nsrow = rows[(ROW0) + t < M ? (ROW0) + t : 0];                     \
  }
#define BWD64_FETCH(ROW0)                                                          \
  {                                                                                \
    if (ROWS) { cid[0] = nid[0]; cid[1] = nid[1]; srow = nsrow; }                  \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                \
      int64_t row = (ROW0) + rsub * 2 + p;                                         \
      const bool in = row < M;                                                     \
      row = in ? (ROWS ? (int64_t)cid[p] : row) : 0;                               \
      gv[p] = *(const float4 *)(g + row * 64 + c4);                                \
      if (MODE != 0) yv[p] = *(const float4 *)(y + row * 64 + c4);                 \
      if (MODE == 2) rn[p] = rnorm[row];                                           \
      av[p] = *(const float4 *)(agg + row * 64 + c4);                              \
      xv[p] = *(const float4 *)(x + row * 64 + c4);                                \
      if (!in) {                                                                   \
        gv[p] = make_float4(0.f, 0.f, 0.f, 0.f);                                   \
        yv[p] = gv[p]; av[p] = gv[p]; xv[p] = gv[p];                               \
        rn[p] = 0.f;                                                               \
      }                                                                            \
    }                                                                              \
    if (t < TR)                                                                    \
      scv = (agg_scale && (ROW0) + t < M) ? agg_scale[ROWS ? (int64_t)srow : (ROW0) + t] : 1.f; \
  }
  // ---- dpre tile and [agg | x] tile of the operands in flight into LDS buffer B_
#define BWD64_STAGE(B_)                                                                              \
  {                                                                                                  \
    float *Dq_ = Dp[B_], *Aq_ = Ac[B_];                                                              \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                  \
      float4 d = gv[p];                                                                              \
      if (MODE == 2) {                                                                               \
        float dot = gv[p].x * yv[p].x + gv[p].y * yv[p].y + gv[p].z * yv[p].z + gv[p].w * yv[p].w;   \
        dot = row16_sum(dot);                                                                        \
        if (rn[p] >= 0.5f / L2_EPS) dot = 0.f;   /* clamped norm: dy/dx = I / eps */                 \
        d.x = (gv[p].x - yv[p].x * dot) * rn[p]; d.y = (gv[p].y - yv[p].y * dot) * rn[p];            \
        d.z = (gv[p].z - yv[p].z * dot) * rn[p]; d.w = (gv[p].w - yv[p].w * dot) * rn[p];            \
      } else if (MODE == 1) {                                                                        \
        d.x = yv[p].x > 0.f ? d.x * inv_keep : 0.f; d.y = yv[p].y > 0.f ? d.y * inv_keep : 0.f;      \
        d.z = yv[p].z > 0.f ? d.z * inv_keep : 0.f; d.w = yv[p].w > 0.f ? d.w * inv_keep : 0.f;      \
      }                                                                                              \
      const int r = rsub * 2 + p;   /* wave w stages rows 8w .. 8w+7 of the tile */                  \
      if (HDX) {                                                                                     \
        /* the row's largest magnitude (its 64 values sit in 16 consecutive lanes) -> its scale */   \
        float m = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));               \
        m = row16_max(m);                                                                            \
        if ((t & 15) == 0) {                                                                         \
          const float sr = P64H_TOP / p64h_pow2_floor(m, P64H_TOP);   /* a power of two (1: nothing usable) */ \
          rsA[B_][r] = sr;                                                                           \
          rsI[B_][r] = __uint_as_float(0x7F000000u - __float_as_uint(sr)) * inv_sw;   /* 1 / sr, exactly */ \
        }                                                                                            \
      }                                                                                              \
      *(float4 *)(Dq_ + r * S64_DP + c4) = d;                                                        \
      *(float4 *)(Aq_ + r * S64_AP + c4) = av[p];                                                    \
      *(float4 *)(Aq_ + r * S64_AP + 64 + c4) = xv[p];                                               \
    }                                                                                                \
    if (t < TR) {                                                                                    \
      sc[B_][t] = scv;                                                                               \
      if (ROWS) rid[B_][t] = srow;                                                                   \
    }                                                                                                \
  }
  const int64_t gstep = gridDim.x;
  if ((int64_t)blockIdx.x < ntiles) {
    BWD64_IDS((int64_t)blockIdx.x * TR)
    BWD64_FETCH((int64_t)blockIdx.x * TR)
    if ((int64_t)blockIdx.x + gstep < ntiles) BWD64_IDS(((int64_t)blockIdx.x + gstep) * TR)
  }
  int buf = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int64_t row0 = tile * TR;
    float *Dpb = Dp[buf], *Acb = Ac[buf];
    BWD64_STAGE(buf)
    // the only barrier of the tile: the other buffer was last read before every wave reached
    // the previous one, so the next tile may be staged into it without a second barrier
    __syncthreads();
    if (tile + gstep < ntiles) {
      BWD64_FETCH((tile + gstep) * TR)
      if (tile + 2 * gstep < ntiles) BWD64_IDS((tile + 2 * gstep) * TR)
    }
    // ---- dX (K = 64 dpre columns) and dW (K = the tile's 32 rows), interleaved
    f32x16 a0;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = 0.f;
    if (HDX) {
      // dX on the fp16 pipe: row lane & 31 of dpre, k = 16 s + 8 (lane >> 5) .. + 7, scaled and cut in two.
      // The four k-steps are spread over the sixteen dW steps (whose LDS operands are fetched one
      // step ahead), so that the splits and the LDS reads run under the fp32 MFMAs.
      const float sr = rsA[buf][lane & 31];
      const float *drow8 = Dpb + (lane & 31) * S64_DP + (lane >> 5) * 8;
      float d0 = Dpb[(lane >> 5) * S64_DP + (lane & 31)];
      float d1 = Dpb[(lane >> 5) * S64_DP + 32 + (lane & 31)];
      float bq = Acb[(lane >> 5) * S64_AP + w * 32 + (lane & 31)];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        float n0 = 0.f, n1 = 0.f, nb = 0.f;
        if (kk < 15) {
          const int kr = 2 * (kk + 1) + (lane >> 5);
          n0 = Dpb[kr * S64_DP + (lane & 31)];
          n1 = Dpb[kr * S64_DP + 32 + (lane & 31)];
          nb = Acb[kr * S64_AP + w * 32 + (lane & 31)];
        }
        dw0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, bq, dw0, 0, 0, 0);
        dw1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, bq, dw1, 0, 0, 0);
        if ((kk & 3) == 0) {
          const int s2 = kk >> 2;
          const float4 f0 = *(const float4 *)(drow8 + 16 * s2), f1 = *(const float4 *)(drow8 + 16 * s2 + 4);
          p64h_f8 a1, a2;
#pragma unroll
          for (int e2 = 0; e2 < 8; e2 += 2) {
            const float v0 = (e2 < 4 ? f4c(f0, e2) : f4c(f1, e2 - 4)) * sr;
            const float v1 = (e2 < 4 ? f4c(f0, e2 + 1) : f4c(f1, e2 - 3)) * sr;
            const p64h_f2 hi = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v0, v1));
            const p64h_f2 lo = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v0 - (float)hi[0], v1 - (float)hi[1]));
            a1[e2] = hi[0]; a1[e2 + 1] = hi[1];
            a2[e2] = lo[0]; a2[e2 + 1] = lo[1];
          }
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, bh[s2], a0, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bl[s2], a0, 0, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh[s2], a0, 0, 0, 0);
        }
        d0 = n0; d1 = n1; bq = nb;
      }
    } else if (!HDX) {
    const float *drow = Dpb + (lane & 31) * S64_DP + (lane >> 5) * 4;
#pragma unroll
    for (int gi = 0; gi < 8; ++gi) {
      const float4 af = *(const float4 *)(drow + gi * 8);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(af, s2), f4c(breg[gi], s2), a0, 0, 0, 0);
        if (s2 & 1) {
          const int kr = 2 * (gi * 2 + (s2 >> 1)) + (lane >> 5);
          const float d0 = Dpb[kr * S64_DP + (lane & 31)];
          const float d1 = Dpb[kr * S64_DP + 32 + (lane & 31)];
          const float b = Acb[kr * S64_AP + w * 32 + (lane & 31)];
          dw0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, b, dw0, 0, 0, 0);
          dw1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, b, dw1, 0, 0, 0);
        }
      }
    }
    }
    if (t < 64) {
      // wave 0 walks the dpre tile for the bias gradient anyway: lane = column, so one ballot per
      // row says whether the row is all zero.  A dpre row of zeros gives rows of zeros in grad_agg
      // and grad_x; one bit per row (a tile of 32 rows = one word) tells the transposed gather
      // which rows it need not fetch.
      unsigned word = 0u;
#pragma unroll 8
      for (int k = 0; k < TR; ++k) {
        const float v = Dpb[k * S64_DP + t];
        bsum += v;
        if (!ROWS) word |= (__ballot(v != 0.f) != 0ull ? 1u : 0u) << k;
      }
      if (!ROWS && nz_mask && t == 0) nz_mask[tile] = word;
    }
    // ---- dX stores straight from the accumulators (128-byte row segments)
    {
      float *dst = (w < 2 ? grad_agg : grad_x) + (w & 1) * 32 + (lane & 31);
      unsigned gamx = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t grow = ROWS ? (int64_t)rid[buf][lr] : row0 + lr;
        float s = w < 2 ? sc[buf][lr] : 1.f;
        if (HDX) s *= rsI[buf][lr];
        if (row0 + lr < M) {
          const float v = a0[r] * s;
          dst[grow * 64] = v;
          gamx = max(gamx, __float_as_uint(v) & 0x7FFFFFFFu);
        }
      }
      if (gmax) gmx[t] = max(gmx[t], gamx);
    }
  }
  if (gmax) amax_publish(gmx[t], gmax);
#undef BWD64_FETCH
#undef BWD64_IDS
#undef BWD64_STAGE
  float *slab = slabs + (int64_t)blockIdx.x * (64 * 128 + 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int mrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int ncol = w * 32 + (lane & 31);
    slab[mrow * 128 + ncol] = dw0[r];
    slab[(32 + mrow) * 128 + ncol] = dw1[r];
  }
  if (t < 64) slab[64 * 128 + t] = bsum;
}

// ---- The row-list backward of a 64 -> 64 layer with BOTH products on the fp16 pipe (round 3, last).
// bwd64_fused_k<.., HDX> cuts its dpre tile in two fp16 parts as each wave reads it back from LDS --
// all four waves need all of it, so every value is cut four times -- and keeps dW on fp32 MFMAs
// (64 cycles each); moving dW to fp16 the same way cost more in cuts and strided LDS reads than its
// MFMAs had (ablation: 0.61 ms of reads + cuts for 0.15 ms of MFMAs).  Here every value is cut ONCE,
// by the thread that stages it, and LDS holds fp16 planes only:
//   Dr  dpre row-major, each row scaled by its own power of two      -> the A operand of dX
//   Dt  dpre transposed ([column][row pair] words), scaled per 8-row block -> the A operand of dW
//   At  [agg | x] transposed the same way                            -> the B operand of dW
// The transposed planes come for free from the cut itself: a thread holds two consecutive rows,
// and v_cvt_pkrtz of (row 2i, row 2i+1) of one column IS the word of that column's row pair.  dW sums
// over rows, so rows that meet in one MFMA need one scale: the 8 rows a wave stages are exactly the
// k = 8 of v_mfma_f32_32x32x8_f16 (no barrier between finding their maximum and cutting them); each
// block's product lands in a fresh accumulator and is added to the running fp32 sums with its own
// exact power of two (chains of 3 MFMAs: nothing for the pipe's truncating accumulate to add up).
// Single LDS stage, two barriers per tile, 38 KB per workgroup.  Anatomy at C4 (4.36 M listed rows, kernel
// 1.54 ms against 1.68 for bwd64_fused_k<.., HDX>): fetch + cut + planes 0.81 ms (4.5 GB of listed rows:
// 5.5 TB/s), the dX phase 0.57 ms -- nearly all of it its 2.2 GB of stores, whole 128-byte lines of
// scattered rows at 3.1 TB/s, which add to the reads instead of hiding under them -- and dW 0.16 ms, of
// which 0.03 are its 24 MFMAs and the rest the eight accumulator flushes per tile (v_pk_fma_f32).
// 6.7 GB in 1.54 ms = 4.35 TB/s: what is left is the memory system's mixed read/write rate on rows
// fetched by number.
typedef _Float16 p64h_f4 __attribute__((ext_vector_type(4)));
typedef short p64h_b4 __attribute__((ext_vector_type(4)));
constexpr int B64H_DRP = 72;   // halves per row of Dr (144 bytes: conflict-free 16-byte row slices)
constexpr int B64H_TP = 18;    // words per column of Dt / At (16 row pairs + 2: conflict-free 8-byte slices)

__device__ __forceinline__ float p64h_inv_pow2(float s) { return __uint_as_float(0x7F000000u - __float_as_uint(s)); }

template <int MODE>   // 0 = no epilogue, 1 = ReLU(+dropout), 2 = L2 normalisation
__global__ void __launch_bounds__(256, 3)
bwd64h_rows_k(const float *__restrict__ g, const float *__restrict__ y, const float *__restrict__ rnorm,
              const float *__restrict__ agg, const float *__restrict__ x,
              const float *__restrict__ agg_scale, float inv_keep, float *__restrict__ grad_agg,
              float *__restrict__ grad_x, float *__restrict__ slabs, const int *__restrict__ rows,
              const int *__restrict__ n_rows, int64_t rows_cap, unsigned *__restrict__ gmax,
              const uint16_t *__restrict__ hplanes, const float *__restrict__ hparams) {
  constexpr int TR = 32;
  __shared__ __align__(16) uint16_t Dr[2 * TR * B64H_DRP];
  __shared__ __align__(16) uint32_t Dt[2 * 64 * B64H_TP];
  __shared__ __align__(16) uint32_t At[2 * 128 * B64H_TP];
  __shared__ float sc[TR], rsI[TR], ubl[4];
  __shared__ int rid[TR];
  __shared__ unsigned gmx[256];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int64_t M = *n_rows;
  if (M > rows_cap) M = rows_cap;
  // dX: wave w owns output columns 32w..32w+31 of [grad_agg | grad_x]; B operand of k-step s:
  // column 32 w + (lane & 31), k = 16 s + 8 (lane >> 5) .. + 7 of both weight planes (bwd64h_pack_k)
  p64h_f8 bh[4], bl[4];
  const float inv_sw = hparams[0];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    const int off = (w * 32 + (lane & 31)) * 64 + 16 * s2 + 8 * (lane >> 5);
    bh[s2] = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(hplanes + off));
    bl[s2] = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(hplanes + 128 * 64 + off));
  }
  f32x16 dw0, dw1;   // dW: rows m = dpre column 0..31 / 32..63, columns 32w .. 32w+31 of [agg | x]
#pragma unroll
  for (int r = 0; r < 16; ++r) { dw0[r] = 0.f; dw1[r] = 0.f; }
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);   // this thread's share of the bias gradient (its 4 columns)
  gmx[t] = 0u;
  const int rsub = t >> 4, c4 = (t & 15) * 4;
  const int wi = rsub;                           // row pair (rows 2 rsub, 2 rsub + 1): word index in Dt / At
  const int64_t ntiles = (M + TR - 1) / TR;
  float4 gv[2], yv[2], av[2], xv[2];
  float rn[2], scv = 1.f;
  int nid[2] = {0, 0}, cid[2] = {0, 0}, srow = 0, nsrow = 0;
#define B64H_IDS(ROW0)                                                             \
  {                                                                                \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                \
      const int64_t pos = (ROW0) + rsub * 2 + p;                                   \
      nid[p] = rows[pos < M ? pos : 0];                                            \
    }                                                                              \
    if (t < TR) nsrow = rows[(ROW0) + t < M ? (ROW0) + t : 0];                     \
  }
#define B64H_FETCH(ROW0)                                                           \
  {                                                                                \
    cid[0] = nid[0]; cid[1] = nid[1]; srow = nsrow;                                \
    _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                \
      const bool in = (ROW0) + rsub * 2 + p < M;                                   \
      const int64_t row = in ? (int64_t)cid[p] : 0;                                \
      gv[p] = *(const float4 *)(g + row * 64 + c4);                                \
      if (MODE != 0) yv[p] = *(const float4 *)(y + row * 64 + c4);                 \
      if (MODE == 2) rn[p] = rnorm[row];                                           \
      av[p] = *(const float4 *)(agg + row * 64 + c4);                              \
      xv[p] = *(const float4 *)(x + row * 64 + c4);                                \
      if (!in) {                                                                   \
        gv[p] = make_float4(0.f, 0.f, 0.f, 0.f);                                   \
        yv[p] = gv[p]; av[p] = gv[p]; xv[p] = gv[p];                               \
        rn[p] = 0.f;                                                               \
      }                                                                            \
    }                                                                              \
    if (t < TR) scv = (agg_scale && (ROW0) + t < M) ? agg_scale[(int64_t)srow] : 1.f; \
  }
  // two values, scaled -> (word of fp16 high parts, word of fp16 low parts), round to zero
#define B64H_CUT(V0, V1, S, HI, LO)                                                                    \
  {                                                                                                    \
    const float v0_ = (V0) * (S), v1_ = (V1) * (S);                                                    \
    const p64h_f2 h_ = __builtin_bit_cast(p64h_f2, __builtin_amdgcn_cvt_pkrtz(v0_, v1_));              \
    HI = __builtin_bit_cast(uint32_t, h_);                                                             \
    LO = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(v0_ - (float)h_[0], v1_ - (float)h_[1])); \
  }
  const int64_t gstep = gridDim.x;
  if ((int64_t)blockIdx.x < ntiles) {
    B64H_IDS((int64_t)blockIdx.x * TR)
    B64H_FETCH((int64_t)blockIdx.x * TR)
    if ((int64_t)blockIdx.x + gstep < ntiles) B64H_IDS(((int64_t)blockIdx.x + gstep) * TR)
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    // ---- the tile's operands, cut once, into the planes
    {
      float4 d[2];
      float md = 0.f, mx = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        d[p] = gv[p];
        if (MODE == 2) {
          float dot = gv[p].x * yv[p].x + gv[p].y * yv[p].y + gv[p].z * yv[p].z + gv[p].w * yv[p].w;
          dot = row16_sum(dot);
          if (rn[p] >= 0.5f / L2_EPS) dot = 0.f;   // clamped norm: dy/dx = I / eps
          d[p].x = (gv[p].x - yv[p].x * dot) * rn[p]; d[p].y = (gv[p].y - yv[p].y * dot) * rn[p];
          d[p].z = (gv[p].z - yv[p].z * dot) * rn[p]; d[p].w = (gv[p].w - yv[p].w * dot) * rn[p];
        } else if (MODE == 1) {
          d[p].x = yv[p].x > 0.f ? d[p].x * inv_keep : 0.f; d[p].y = yv[p].y > 0.f ? d[p].y * inv_keep : 0.f;
          d[p].z = yv[p].z > 0.f ? d[p].z * inv_keep : 0.f; d[p].w = yv[p].w > 0.f ? d[p].w * inv_keep : 0.f;
        }
        bs.x += d[p].x; bs.y += d[p].y; bs.z += d[p].z; bs.w += d[p].w;
        const int r = rsub * 2 + p;
        // the row's largest magnitude (its 64 values sit in 16 consecutive lanes) -> its scale for dX
        float m = fmaxf(fmaxf(fabsf(d[p].x), fabsf(d[p].y)), fmaxf(fabsf(d[p].z), fabsf(d[p].w)));
        m = row16_max(m);
        md = fmaxf(md, m);
        const float sr = P64H_TOP / p64h_pow2_floor(m, P64H_TOP);   // a power of two (1: nothing usable)
        if ((t & 15) == 0) rsI[r] = p64h_inv_pow2(sr) * inv_sw;
        uint2 hq, lq;
        B64H_CUT(d[p].x, d[p].y, sr, hq.x, lq.x)
        B64H_CUT(d[p].z, d[p].w, sr, hq.y, lq.y)
        *(uint2 *)(Dr + r * B64H_DRP + c4) = hq;
        *(uint2 *)(Dr + TR * B64H_DRP + r * B64H_DRP + c4) = lq;
        float ma = fmaxf(fmaxf(fmaxf(fabsf(av[p].x), fabsf(av[p].y)), fmaxf(fabsf(av[p].z), fabsf(av[p].w))),
                         fmaxf(fmaxf(fabsf(xv[p].x), fabsf(xv[p].y)), fmaxf(fabsf(xv[p].z), fabsf(xv[p].w))));
        mx = fmaxf(mx, row16_max(ma));
      }
      // the wave's 8 rows = one k-block of dW: their common scales (lanes 16 apart hold other row pairs)
      md = fmaxf(md, __shfl_xor(md, 16, 64)); md = fmaxf(md, __shfl_xor(md, 32, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float sd = P64H_TOP / p64h_pow2_floor(md, P64H_TOP), sx = P64H_TOP / p64h_pow2_floor(mx, P64H_TOP);
      if (lane == 0) ubl[w] = p64h_inv_pow2(sd) * p64h_inv_pow2(sx);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint32_t hi, lo;
        B64H_CUT(f4c(d[0], i), f4c(d[1], i), sd, hi, lo)
        Dt[(c4 + i) * B64H_TP + wi] = hi;
        Dt[64 * B64H_TP + (c4 + i) * B64H_TP + wi] = lo;
        B64H_CUT(f4c(av[0], i), f4c(av[1], i), sx, hi, lo)
        At[(c4 + i) * B64H_TP + wi] = hi;
        At[128 * B64H_TP + (c4 + i) * B64H_TP + wi] = lo;
        B64H_CUT(f4c(xv[0], i), f4c(xv[1], i), sx, hi, lo)
        At[(64 + c4 + i) * B64H_TP + wi] = hi;
        At[128 * B64H_TP + (64 + c4 + i) * B64H_TP + wi] = lo;
      }
      if (t < TR) {
        sc[t] = scv;
        rid[t] = srow;
      }
    }
    __syncthreads();
    if (tile + gstep < ntiles) {
      B64H_FETCH((tile + gstep) * TR)
      if (tile + 2 * gstep < ntiles) B64H_IDS((tile + 2 * gstep) * TR)
    }
    // ---- dX = dpre . [W_l | W_r]: row lane & 31, k = 16 s + 8 (lane >> 5) .. + 7
    {
      f32x16 a0;
#pragma unroll
      for (int r = 0; r < 16; ++r) a0[r] = 0.f;
      const uint16_t *dr = Dr + (lane & 31) * B64H_DRP + 8 * (lane >> 5);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const p64h_f8 ah = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(dr + 16 * s2));
        const p64h_f8 al = __builtin_bit_cast(p64h_f8, *(const p64h_b8 *)(dr + TR * B64H_DRP + 16 * s2));
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s2], a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s2], a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s2], a0, 0, 0, 0);
      }
      float *dst = (w < 2 ? grad_agg : grad_x) + (w & 1) * 32 + (lane & 31);
      unsigned gamx = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t grow = (int64_t)rid[lr];
        const float s = (w < 2 ? sc[lr] : 1.f) * rsI[lr];
        if (row0 + lr < M) {
          const float v = a0[r] * s;
          dst[grow * 64] = v;
          gamx = max(gamx, __float_as_uint(v) & 0x7FFFFFFFu);
        }
      }
      if (gmax) gmx[t] = max(gmx[t], gamx);
    }
    // ---- dW[m = dpre column][n = column 32 w + .. of [agg | x]] += block by block (8 rows each)
    {
      const uint32_t *atp = At + (w * 32 + (lane & 31)) * B64H_TP + 2 * (lane >> 5);
      const uint32_t *dtp = Dt + (lane & 31) * B64H_TP + 2 * (lane >> 5);
      f32x16 zero;
#pragma unroll
      for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll 1     // (unrolled, the eight independent products are scheduled side by side: 128 registers)
      for (int b = 0; b < 4; ++b) {
        const float un = ubl[b];
        f32x16 unv;
#pragma unroll
        for (int r = 0; r < 16; ++r) unv[r] = un;
        const p64h_f4 xh = __builtin_bit_cast(p64h_f4, *(const p64h_b4 *)(atp + 4 * b));
        const p64h_f4 xl = __builtin_bit_cast(p64h_f4, *(const p64h_b4 *)(atp + 128 * B64H_TP + 4 * b));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const p64h_f4 dh = __builtin_bit_cast(p64h_f4, *(const p64h_b4 *)(dtp + 32 * h * B64H_TP + 4 * b));
          const p64h_f4 dl = __builtin_bit_cast(p64h_f4, *(const p64h_b4 *)(dtp + (64 + 32 * h) * B64H_TP + 4 * b));
          f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x8f16(dl, xh, zero, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x8f16(dh, xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x8f16(dh, xh, acc, 0, 0, 0);
          // (whole-vector form: v_pk_fma_f32, two sums per instruction)
          if (h == 0) dw0 = __builtin_elementwise_fma(acc, unv, dw0);
          else dw1 = __builtin_elementwise_fma(acc, unv, dw1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __syncthreads();     // every wave is done with the planes: the next tile may overwrite them
  }
  if (gmax) amax_publish(gmx[t], gmax);
#undef B64H_IDS
#undef B64H_FETCH
#undef B64H_CUT
  float *slab = slabs + (int64_t)blockIdx.x * (64 * 128 + 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mrow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int ncol = w * 32 + (lane & 31);
    slab[mrow * 128 + ncol] = dw0[r];
    slab[(32 + mrow) * 128 + ncol] = dw1[r];
  }
  // bias gradient: the 16 threads that staged a column quad, summed in a fixed order
  float *bred = (float *)At;
  *(float4 *)(bred + rsub * 64 + c4) = bs;
  __syncthreads();
  if (t < 64) {
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += bred[k * 64 + t];
    slab[64 * 128 + t] = sum;
  }
}

constexpr int S64_GRID = 256 * 3;   // persistent workgroups (3 per CU fit LDS)
constexpr int S64_BWD_GRID = 256 * 3;

// =============================================================================== VALU
// small-K forward (2*Fi <= 16, Fo <= 64): wave per row, lane = output column, weights in
// registers, the row's 2*Fi inputs fetched by one load and broadcast with v_readlane.
template <int K2MAX>
__global__ void __launch_bounds__(256)
project_fwd_smallk_k(const float *__restrict__ agg, const float *__restrict__ x,
                     const float *__restrict__ wt, int64_t N, int Fi, int Fo, EpiArgs e,
                     float *__restrict__ out) {
  const int lane = gs_lane();
  const int K2 = 2 * Fi;
  const int col = blockIdx.y * 64 + lane;   // blockIdx.y: block of 64 output columns
  float wreg[K2MAX];
#pragma unroll
  for (int k = 0; k < K2MAX; ++k) wreg[k] = (k < K2 && col < Fo) ? wt[k * Fo + col] : 0.f;
  const float bias = col < Fo ? e.bias[col] : 0.f;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < N; row += nwaves) {
    float in = 0.f;
    if (lane < Fi) in = agg[row * Fi + lane];
    else if (lane < K2) in = x[row * Fi + lane - Fi];
    float v = bias;
#pragma unroll
    for (int k = 0; k < K2MAX; ++k) {
      float ak = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, in), k));
      v = fmaf(ak, wreg[k], v);
    }
    if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) v = fmaxf(v, 0.f);
    if (e.mode == GSAGE_EPI_RELU_DROPOUT && col < Fo) {
      uint64_t idx = (uint64_t)row * (uint64_t)Fo + (uint64_t)col;
      bool keep = e.keep_mask ? (e.keep_mask[idx] != 0) : dropout_keep(e.seed, idx, e.p);
      v = keep ? v * e.inv_keep : 0.f;
    }
    if (e.mode == GSAGE_EPI_L2NORM) {   // only launched with one column block (Fo <= 64)
      float ss = gs_wave_sum(col < Fo ? v * v : 0.f);
      float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
      v *= rn;
      if (lane == 0 && e.rnorm) e.rnorm[row] = rn;
    }
    if (col < Fo) out[row * Fo + col] = v;
  }
}


// first-layer forward on the matrix cores: Fi = 8 (the 7 memory features padded), so
// K = [agg | x] = 16.  Computed transposed, D^T[n][row] = sum_k W^T[n][k] * cat[row][k], with
// v_mfma_f32_16x16x4_f32: lane (row = l&15, q = l>>4) loads ONE float4 of its row
// (q = 0,1: agg columns 0-3 / 4-7; q = 2,3: x columns) and ends up holding 4 consecutive output
// columns of that row per 16-column tile, so outputs leave as 16-byte stores.  The weights sit
// in 16 registers per lane for the whole launch.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
project_fwd_k16_k(const float *__restrict__ agg, const float *__restrict__ x,
                  const float *__restrict__ wt /* [16][Fo] */, int64_t N, int Fo, EpiArgs e,
                  float *__restrict__ out, uint64_t *__restrict__ sign, int64_t out_blk = 0,
                  float *__restrict__ U = nullptr, int urow = 16) {
  // U (optional): the rows u[j] = [agg[j][0..6], m_lo, x[j][0..6], m_hi] that the recompute gather
  // of the NEXT layer fetches (hidden_gather.hip; `urow` floats per row: 16, or 32 when the masks
  // of further 64-column groups ride behind the row).  m = the 64 keep-and-positive bits of this
  // workgroup's column group, in the order the ballots below come in; this kernel already holds
  // the row's operands and those ballots, so no separate pack pass over agg / x / sign is needed.
  constexpr int K16_PP = 68;                   // pitch of the 16 x 64 output patch
  __shared__ __align__(16) float patches[4][16 * K16_PP];
  float *patch = patches[threadIdx.x >> 6];
  const int lane = gs_lane();
  const int rl = lane & 15, q = lane >> 4;
  const int cb = blockIdx.y * 64;              // this workgroup's 64 output columns
  // A operand of step s, tile nt: W^T[n = cb+16nt+rl][k' = 4q+s]
  float wreg[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) wreg[nt][s2] = wt[(4 * q + s2) * Fo + cb + 16 * nt + rl];
  float4 bias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bias[nt] = *(const float4 *)(e.bias + cb + 16 * nt + 4 * q);
  const float *src = q < 2 ? agg + 4 * q : x + 4 * (q - 2);
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  unsigned amx = 0u, uamx = 0u;
  // every element index of the launch below 2^32 (uniform): the four-at-a-time form of the dropout hash
  const bool idx32 = (uint64_t)N * (uint64_t)Fo <= 0x100000000ull;
  const uint32_t thr = dropout_threshold(e.p);
  for (int64_t r0 = wave0 * 16; r0 < N; r0 += nwaves * 16) {
    const int64_t row = r0 + rl;
    const bool live = row < N;
    float4 c = live ? *(const float4 *)(src + row * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (U) {     // max |u| over the table, published behind it for the gather's fp16 scaling
      uamx = max(max(uamx, __float_as_uint(c.x) & 0x7FFFFFFFu), __float_as_uint(c.y) & 0x7FFFFFFFu);
      uamx = max(max(uamx, __float_as_uint(c.z) & 0x7FFFFFFFu), __float_as_uint(c.w) & 0x7FFFFFFFu);
    }
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][0], c.x, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][1], c.y, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][2], c.z, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[nt][3], c.w, acc[nt], 0, 0, 0);
    }
    // acc[nt][reg] = out[row][cb + 16nt + 4q + reg]
    float4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      v[nt] = make_float4(acc[nt][0] + bias[nt].x, acc[nt][1] + bias[nt].y,
                          acc[nt][2] + bias[nt].z, acc[nt][3] + bias[nt].w);
      if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) {
        v[nt].x = fmaxf(v[nt].x, 0.f); v[nt].y = fmaxf(v[nt].y, 0.f);
        v[nt].z = fmaxf(v[nt].z, 0.f); v[nt].w = fmaxf(v[nt].w, 0.f);
      }
      if (e.mode == GSAGE_EPI_RELU_DROPOUT && live) {
        uint64_t idx = (uint64_t)row * (uint64_t)Fo + (uint64_t)(cb + 16 * nt + 4 * q);
        bool k0, k1, k2, k3;
        if (e.keep_mask) {
          uchar4 m = *(const uchar4 *)(e.keep_mask + idx);
          k0 = m.x; k1 = m.y; k2 = m.z; k3 = m.w;
        } else if (idx32) {
          bool kk[4];
          dropout_keep4_lo(e.seed, (uint32_t)idx, thr, kk);
          k0 = kk[0]; k1 = kk[1]; k2 = kk[2]; k3 = kk[3];
        } else {
          k0 = dropout_keep(e.seed, idx + 0, e.p); k1 = dropout_keep(e.seed, idx + 1, e.p);
          k2 = dropout_keep(e.seed, idx + 2, e.p); k3 = dropout_keep(e.seed, idx + 3, e.p);
        }
        v[nt].x = k0 ? v[nt].x * e.inv_keep : 0.f; v[nt].y = k1 ? v[nt].y * e.inv_keep : 0.f;
        v[nt].z = k2 ? v[nt].z * e.inv_keep : 0.f; v[nt].w = k3 ? v[nt].w * e.inv_keep : 0.f;
      }
      ss += v[nt].x * v[nt].x + v[nt].y * v[nt].y + v[nt].z * v[nt].z + v[nt].w * v[nt].w;
    }
    if (e.mode == GSAGE_EPI_L2NORM) {          // launched with one column block only
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { v[nt].x *= rn; v[nt].y *= rn; v[nt].z *= rn; v[nt].w *= rn; }
      if (live && q == 0 && e.rnorm) e.rnorm[row] = rn;
    }
    // a lane holds 16-byte pieces of 4 different 64-byte chunks of its row; stored like that
    // every store instruction writes sixteen 64-byte chunks and the kernel runs at 3.6 TB/s
    // (tools/micro/streambench v4).  Through a wave-private LDS patch each store instruction
    // covers 4 whole 256-byte row segments instead (v4b: 4.9 TB/s).
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) *(float4 *)(patch + rl * K16_PP + 16 * nt + 4 * q) = v[nt];
    uint32_t m_lo = 0u, m_hi = 0u;                    // (U, training) the mask words of row rl
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = 4 * i + q;                       // q = lane >> 4: 4 rows per instruction
      const int64_t orow = r0 + rr;
      const float4 o = *(const float4 *)(patch + rr * K16_PP + rl * 4);
      if (orow < N) {
        // out_blk: the output as Fo/64 separate [N,64] tables (this workgroup writes table cb/64)
        float *dst = out_blk ? out + (int64_t)blockIdx.y * out_blk + orow * 64 + rl * 4
                             : out + orow * Fo + cb + rl * 4;
        *(float4 *)dst = o;
        amx = max(max(amx, __float_as_uint(o.x) & 0x7FFFFFFFu), __float_as_uint(o.y) & 0x7FFFFFFFu);
        amx = max(max(amx, __float_as_uint(o.z) & 0x7FFFFFFFu), __float_as_uint(o.w) & 0x7FFFFFFFu);
      }
      if (sign || (U && e.mode == GSAGE_EPI_RELU_DROPOUT)) {
        // which outputs are positive, for the backward's ReLU'/dropout' (gsage_relu_sign_words):
        // per group of 4 rows and column block four 64-bit words, word c bit l = component c of
        // lane l's float4 -- the lane layout dw16_mfma_k reads the rows back in
        const unsigned long long b0 = __ballot(o.x > 0.f), b1 = __ballot(o.y > 0.f);
        const unsigned long long b2 = __ballot(o.z > 0.f), b3 = __ballot(o.w > 0.f);
        if (sign) {
          const unsigned long long mine = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
          const int64_t grp = (r0 >> 2) + i;
          if (lane < 4 && r0 + 4 * i < N)
            sign[((int64_t)blockIdx.y * ((N + 3) >> 2) + grp) * 4 + lane] = mine;
        }
        if (U && (rl >> 2) == i) {
          // row rl = 4 i + (rl & 3) of the block: its bits are the 16-bit field (rl & 3) of each word
          const int sh = 16 * (rl & 3);
          m_lo = (uint32_t)((b0 >> sh) & 0xFFFFull) | ((uint32_t)((b1 >> sh) & 0xFFFFull) << 16);
          m_hi = (uint32_t)((b2 >> sh) & 0xFFFFull) | ((uint32_t)((b3 >> sh) & 0xFFFFull) << 16);
        }
      }
    }
    if (U && live) {
      float *u = U + row * urow;
      if (blockIdx.y == 0) {
        // lane (rl, q) holds quarter q of [agg | x] of row rl: 16 lanes x 16 bytes = the 64-byte row,
        // a wave = 16 consecutive rows = 1 KiB contiguous (2 KiB-strided halves when urow = 32)
        float4 cu = c;
        if (q == 1) cu.w = __uint_as_float(m_lo);
        if (q == 3) cu.w = __uint_as_float(m_hi);
        *(float4 *)(u + 4 * q) = cu;
      } else if (urow > 16) {
        // column group g >= 1: its two mask words behind the row (low words at float 16, high at 20)
        if (q == 1) ((uint32_t *)u)[16 + blockIdx.y - 1] = m_lo;
        if (q == 3) ((uint32_t *)u)[20 + blockIdx.y - 1] = m_hi;
      }
    }
  }
  if (e.amax_out) amax_publish(amx, e.amax_out);
  if (U && blockIdx.y == 0) amax_publish(uamx, (unsigned *)(U + N * urow));
}

// the word behind the sign bits says whether the forward kernel that ran filled them in
__global__ void sign_flag_k(uint64_t *flag, uint64_t v) { *flag = v; }

// generic forward: wave per row, lanes stride the output columns; Wt from L1/L2.
__global__ void __launch_bounds__(256)
project_fwd_generic_k(const float *__restrict__ agg, const float *__restrict__ x,
                      const float *__restrict__ wt, int64_t N, int Fi, int Fo, EpiArgs e,
                      float *out) {
  const int lane = gs_lane();
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < N; row += nwaves) {
    const float *ar = agg + row * Fi, *xr = x + row * Fi;
    float ss = 0.f;
    for (int j = lane; j < Fo; j += 64) {
      float v = e.bias[j];
      for (int k = 0; k < Fi; ++k) v = fmaf(ar[k], wt[k * Fo + j], v);
      for (int k = 0; k < Fi; ++k) v = fmaf(xr[k], wt[(Fi + k) * Fo + j], v);
      if (e.mode == GSAGE_EPI_RELU || e.mode == GSAGE_EPI_RELU_DROPOUT) v = fmaxf(v, 0.f);
      if (e.mode == GSAGE_EPI_RELU_DROPOUT) {
        uint64_t idx = (uint64_t)row * (uint64_t)Fo + (uint64_t)j;
        bool keep = e.keep_mask ? (e.keep_mask[idx] != 0) : dropout_keep(e.seed, idx, e.p);
        v = keep ? v * e.inv_keep : 0.f;
      }
      ss += v * v;
      out[row * Fo + j] = v;
    }
    if (e.mode == GSAGE_EPI_L2NORM) {
      ss = gs_wave_sum(ss);
      float rn = 1.0f / fmaxf(sqrtf(ss), L2_EPS);
      for (int j = lane; j < Fo; j += 64) out[row * Fo + j] *= rn;  // same lane wrote it
      if (lane == 0 && e.rnorm) e.rnorm[row] = rn;
    }
  }
}

// ------------------------------------------------------------------- epilogue backward
// dpre = d loss / d (pre-epilogue activations); wave per row.
__global__ void __launch_bounds__(256)
epilogue_bwd_k(const float *__restrict__ g, const float *__restrict__ out,
               const float *__restrict__ rnorm, int64_t N, int Fo, int mode, float inv_keep,
               float *__restrict__ dpre) {
  const int lane = gs_lane();
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < N; row += nwaves) {
    const float *gr = g + row * Fo, *yr = out + row * Fo;
    float *dr = dpre + row * Fo;
    if (mode == GSAGE_EPI_L2NORM) {
      float rn = rnorm[row];
      if (rn >= 0.5f / L2_EPS) {  // norm was clamped to eps: y = x/eps, dy/dx = I/eps
        for (int j = lane; j < Fo; j += 64) dr[j] = gr[j] * rn;
      } else {
        float dot = 0.f;
        for (int j = lane; j < Fo; j += 64) dot = fmaf(gr[j], yr[j], dot);
        dot = gs_wave_sum(dot);
        for (int j = lane; j < Fo; j += 64) dr[j] = (gr[j] - yr[j] * dot) * rn;
      }
    } else {  // RELU / RELU_DROPOUT: out > 0 <=> pre > 0 and kept
      for (int j = lane; j < Fo; j += 64) dr[j] = yr[j] > 0.f ? gr[j] * inv_keep : 0.f;
    }
  }
}

// generic dX: wave per row, lanes stride the input columns.
__global__ void __launch_bounds__(256)
dx_generic_k(const float *__restrict__ dpre, const float *__restrict__ wl,
             const float *__restrict__ wr, const float *__restrict__ agg_scale, int64_t N,
             int Fi, int Fo, float *__restrict__ grad_agg, float *__restrict__ grad_x) {
  const int lane = gs_lane();
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave0; row < N; row += nwaves) {
    const float *dr = dpre + row * Fo;
    float s = agg_scale ? agg_scale[row] : 1.f;
    for (int k = lane; k < Fi; k += 64) {
      float ga = 0.f, gx = 0.f;
      for (int j = 0; j < Fo; ++j) {
        float d = dr[j];
        ga = fmaf(d, wl[j * Fi + k], ga);
        gx = fmaf(d, wr[j * Fi + k], gx);
      }
      if (grad_agg) grad_agg[row * Fi + k] = ga * s;
      if (grad_x) grad_x[row * Fi + k] = gx;
    }
  }
}

// small-K dW (2*Fi <= 16, Fo <= 64): lane = output row m of dW, K2MAX accumulators.
template <int K2MAX>
__global__ void __launch_bounds__(256)
dw_smallk_k(const float *__restrict__ dpre, const float *__restrict__ out_saved, int relu_mode,
            float inv_keep, const float *__restrict__ agg, const float *__restrict__ x, int64_t M,
            int Fi, int Fo, int64_t rows_per_block, float *__restrict__ slabs) {
  // relu_mode != 0: `dpre` holds grad_out and the ReLU(+dropout) derivative is applied here
  // from the saved output (out > 0 <=> pre > 0 and kept), saving the dpre round trip
  __shared__ float red[4][K2MAX + 1][64];
  const int lane = gs_lane(), w = threadIdx.x >> 6;
  const int K2 = 2 * Fi;
  int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  float acc[K2MAX + 1];
#pragma unroll
  for (int k = 0; k <= K2MAX; ++k) acc[k] = 0.f;
  for (int64_t row = r_begin + w; row < r_end; row += 4) {
    float d = lane < Fo ? dpre[row * Fo + lane] : 0.f;
    if (relu_mode && lane < Fo) d = out_saved[row * Fo + lane] > 0.f ? d * inv_keep : 0.f;
    float in = 0.f;
    if (lane < Fi) in = agg[row * Fi + lane];
    else if (lane < K2) in = x[row * Fi + lane - Fi];
#pragma unroll
    for (int k = 0; k < K2MAX; ++k) {
      float ak = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, in), k));
      acc[k] = fmaf(d, ak, acc[k]);
    }
    acc[K2MAX] += d;
  }
#pragma unroll
  for (int k = 0; k <= K2MAX; ++k) red[w][k][lane] = acc[k];
  __syncthreads();
  float *slab = slabs + (int64_t)blockIdx.x * ((int64_t)Fo * K2 + Fo);
  for (int e = threadIdx.x; e < (K2MAX + 1) * 64; e += 256) {
    int k = e / 64, m = e % 64;
    float s = red[0][k][m] + red[1][k][m] + red[2][k][m] + red[3][k][m];
    if (m < Fo) {
      if (k < K2) slab[(int64_t)m * K2 + k] = s;
      else if (k == K2MAX) slab[(int64_t)Fo * K2 + m] = s;
    }
  }
}


// first-layer dW on the matrix cores (2*Fi <= 16 columns of [agg | x], Fo = 64):
//   dW[m][k'] += dpre[row][m] * [agg | x][row][k'],  db[m] += dpre[row][m]
// 16 lanes x 16 B cover one 64-float dpre row, so a wave holds 4 rows; those 4 rows are the K
// dimension of v_mfma_f32_16x16x4_f32, float4 component c of a lane is tile c of dpre^T
// (m = 4*sub + c) and the lane's single [agg | x] value is the B operand: 4 MFMAs per 4 rows,
// no cross-lane traffic.  ReLU'(+dropout') is applied on the fly from the saved output.
constexpr int DW16_SLAB = 64 * 16 + 64;   // [m][k'] then bias[m]; k' < 8: agg col, else x col
constexpr int DW16_BLOCKS = 256 * 4;

// the row loop of dw16_mfma_k; BITS: ReLU'/dropout' from the forward's sign bits (one uniform
// 32-byte read per 4 rows) instead of the saved output rows (both loops free of branches around
// their loads)
template <bool BITS>
__device__ __forceinline__ void dw16_rows(const float *__restrict__ g, const float *__restrict__ y,
                                          const uint64_t *__restrict__ sg, int relu_mode,
                                          float inv_keep, const float *__restrict__ agg,
                                          const float *__restrict__ x, int64_t M, int Fi, int Fo,
                                          int64_t wave0, int64_t nwaves, f32x4 (&acc)[4],
                                          float4 &bacc) {
  const int lane = gs_lane();
  const int slot = lane >> 4, sub = lane & 15;
  const int64_t ngroups = (M + 3) >> 2;
  constexpr int U = 4;   // 4 row groups (16 rows) in flight per wave
  for (int64_t r0 = wave0 * (4 * U); r0 < M; r0 += nwaves * (4 * U)) {
    float4 d[U], yv[U];
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t row = r0 + u * 4 + slot;
      bool live = row < M;
      d[u] = live ? *(const float4 *)(g + row * Fo + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (BITS) {
        int64_t grp = (r0 >> 2) + u;
        grp = grp < ngroups ? grp : 0;          // rows past M carry a zero gradient anyway
        const uint64_t *w4 = sg + grp * 4;
        yv[u] = make_float4((float)((w4[0] >> lane) & 1ull), (float)((w4[1] >> lane) & 1ull),
                            (float)((w4[2] >> lane) & 1ull), (float)((w4[3] >> lane) & 1ull));
      } else {
        yv[u] = (live && relu_mode) ? *(const float4 *)(y + row * Fo + sub * 4)
                                    : make_float4(1.f, 1.f, 1.f, 1.f);
      }
      float av = 0.f;
      if (live) {
        if (sub < 8) { if (sub < Fi) av = agg[row * Fi + sub]; }
        else if (sub - 8 < Fi) av = x[row * Fi + sub - 8];
      }
      a[u] = av;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 v = d[u];
      if (relu_mode) {
        v.x = yv[u].x > 0.f ? v.x * inv_keep : 0.f; v.y = yv[u].y > 0.f ? v.y * inv_keep : 0.f;
        v.z = yv[u].z > 0.f ? v.z * inv_keep : 0.f; v.w = yv[u].w > 0.f ? v.w * inv_keep : 0.f;
      }
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, a[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, a[u], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, a[u], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, a[u], acc[3], 0, 0, 0);
      bacc.x += v.x; bacc.y += v.y; bacc.z += v.z; bacc.w += v.w;
    }
  }
}

__global__ void __launch_bounds__(256)
dw16_mfma_k(const float *__restrict__ g, const float *__restrict__ y, int relu_mode,
            float inv_keep, const float *__restrict__ agg, const float *__restrict__ x, int64_t M,
            int Fi, int Fo, float *__restrict__ slabs, const uint64_t *__restrict__ sign) {
  // blockIdx.y selects a block of 64 output columns (Fo = 64, 128, 256 ...)
  g += blockIdx.y * 64;
  y += blockIdx.y * 64;
  __shared__ float red[4][17][64];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  const int slot = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  // the forward's sign bits, when it left them (flag word behind them), replace the 256-byte
  // read of the saved output row by 32 bytes per 4 rows
  const int64_t ngroups = (M + 3) >> 2;
  const bool bits = relu_mode && sign && sign[(int64_t)gridDim.y * ngroups * 4] != 0ull;
  if (bits)
    dw16_rows<true>(g, y, sign + (int64_t)blockIdx.y * ngroups * 4, 1, inv_keep, agg, x, M, Fi, Fo,
                    wave0, nwaves, acc, bacc);
  else
    dw16_rows<false>(g, y, nullptr, relu_mode, inv_keep, agg, x, M, Fi, Fo, wave0, nwaves, acc, bacc);
  bacc.x += __shfl_xor(bacc.x, 16, 64); bacc.y += __shfl_xor(bacc.y, 16, 64);
  bacc.z += __shfl_xor(bacc.z, 16, 64); bacc.w += __shfl_xor(bacc.w, 16, 64);
  bacc.x += __shfl_xor(bacc.x, 32, 64); bacc.y += __shfl_xor(bacc.y, 32, 64);
  bacc.z += __shfl_xor(bacc.z, 32, 64); bacc.w += __shfl_xor(bacc.w, 32, 64);
  // acc[c][reg] of lane l = dW[m = 4*(4*(l>>4)+reg) + c][k' = l & 15]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][c * 4 + r][lane] = acc[c][r];
  red[wave][16][lane] = slot == 0 ? bacc.x : slot == 1 ? bacc.y : slot == 2 ? bacc.z : bacc.w;
  __syncthreads();
  float *slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * DW16_SLAB;
  for (int e = threadIdx.x; e < 17 * 64; e += 256) {
    int q = e >> 6, l = e & 63;
    float v = red[0][q][l] + red[1][q][l] + red[2][q][l] + red[3][q][l];
    if (q < 16) {
      int c = q >> 2, r = q & 3;
      slab[(4 * (4 * (l >> 4) + r) + c) * 16 + (l & 15)] = v;
    } else {
      slab[64 * 16 + 4 * (l & 15) + (l >> 4)] = v;   // lane (slot, sub) holds m = 4*sub + slot
    }
  }
}

// lane = output element (coalesced over a slab's inner index); the 4 waves split the slabs
// and combine through LDS in wave order.
__global__ void __launch_bounds__(256)
reduce_dw16_k(const float *__restrict__ slabs, int nslab, int Fi, float *__restrict__ gwl,
              float *__restrict__ gwr, float *__restrict__ gb) {
  __shared__ float red[4][64];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  slabs += (int64_t)blockIdx.y * nslab * DW16_SLAB;   // this block of 64 output columns
  float s = 0.f;
  if (e < DW16_SLAB)
    for (int b = wave; b < nslab; b += 4) s += slabs[(int64_t)b * DW16_SLAB + e];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && e < DW16_SLAB) {
    float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (e < 64 * 16) {
      int m = blockIdx.y * 64 + (e >> 4), k = e & 15;
      if (k < 8) { if (k < Fi) gwl[m * Fi + k] = v; }
      else if (k - 8 < Fi) gwr[m * Fi + k - 8] = v;
    } else {
      gb[blockIdx.y * 64 + e - 64 * 16] = v;
    }
  }
}

// generic dW: thread per output element, rows of the block's chunk in order.
__global__ void __launch_bounds__(256)
dw_generic_k(const float *__restrict__ dpre, const float *__restrict__ agg,
             const float *__restrict__ x, int64_t M, int Fi, int Fo, int64_t rows_per_block,
             float *__restrict__ slabs) {
  const int K2 = 2 * Fi;
  int64_t r_begin = (int64_t)blockIdx.x * rows_per_block;
  int64_t r_end = r_begin + rows_per_block;
  if (r_end > M) r_end = M;
  float *slab = slabs + (int64_t)blockIdx.x * ((int64_t)Fo * K2 + Fo);
  const int total = Fo * K2 + Fo;
  for (int e = threadIdx.x; e < total; e += 256) {
    float s = 0.f;
    if (e < Fo * K2) {
      int m = e / K2, k = e % K2;
      const float *src = k < Fi ? agg + k : x + (k - Fi);
      for (int64_t row = r_begin; row < r_end; ++row) s = fmaf(dpre[row * Fo + m], src[row * Fi], s);
    } else {
      int m = e - Fo * K2;
      for (int64_t row = r_begin; row < r_end; ++row) s += dpre[row * Fo + m];
    }
    slab[e] = s;
  }
}

// sum the slabs and split into grad_W_l, grad_W_r, grad_b.  lane = output element (coalesced
// over a slab's inner index); the 4 waves take slabs w, w+4, ... and combine through LDS in
// wave order: a fixed order, so the result is bitwise reproducible.
__global__ void __launch_bounds__(256)
reduce_slabs_k(const float *__restrict__ slabs, int nslab, int Fi, int Fo,
               float *__restrict__ gwl, float *__restrict__ gwr, float *__restrict__ gb) {
  __shared__ float red[4][64];
  const int lane = gs_lane(), wave = threadIdx.x >> 6;
  const int K2 = 2 * Fi;
  const int total = Fo * K2 + Fo;
  const int e = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (e < total)
    for (int b = wave; b < nslab; b += 4) s += slabs[(int64_t)b * total + e];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && e < total) {
    float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (e < Fo * K2) {
      int m = e / K2, k = e % K2;
      if (k < Fi) gwl[m * Fi + k] = v;
      else gwr[m * Fi + k - Fi] = v;
    } else {
      gb[e - Fo * K2] = v;
    }
  }
}

bool mfma_dim(int64_t f) { return f == 64 || f == 128 || f == 256; }

int num_slabs(int64_t N) {
  int64_t s = (N + 255) / 256;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

}  // namespace

// weight pack: the fp32 [Ncols][K] matrix, then its three bf16 planes for the split kernels
static size_t pack_f32_bytes(int64_t Fi, int64_t Fo) {
  return gs_align_up((size_t)(2 * Fi * Fo) * sizeof(float), 256);
}
static size_t pack_planes_bytes(int64_t Fi, int64_t Fo) {
  return gs_align_up((size_t)(3 * 2 * Fi * Fo) * sizeof(uint16_t), 256);
}
static size_t pack_bytes(int64_t Fi, int64_t Fo) {   // fp32 pack | bf16/fp16 planes | HalfCtl
  return pack_f32_bytes(Fi, Fo) + pack_planes_bytes(Fi, Fo) + gs_align_up(sizeof(HalfCtl), 256);
}
// GSAGE_WIDE_GEMM=f32 keeps the wide layers on the fp32 MFMA kernels, =split puts every 256-column
// tile on the bf16-split kernel whatever its K (A/B measurements, tests); unset: chosen by shape
static int wide_gemm_mode() {
  static const int v = [] {
    const char *e = getenv("GSAGE_WIDE_GEMM");
    if (!e) return -1;
    return e[0] == 'f' ? 0 : e[0] == 's' ? 1 : e[0] == 'h' ? 2 : -1;
  }();
  return v;
}
// GSAGE_PROJ64=f32 keeps the 64 -> 64 forward on the fp32 MFMA kernel even when a bound on |x| is
// handed in (A/B measurements)
static bool proj64_half_mode() {
  static const bool v = [] {
    const char *e = getenv("GSAGE_PROJ64");
    return !(e && e[0] == 'f');
  }();
  return v;
}
// the fused 64 -> 64 backward, row-list form: GSAGE_BWD64=f32 keeps it on fp32 MFMAs, =dx puts only
// its dX half on the fp16 pipe (bwd64_fused_k<.., HDX>); default: dX and dW both (bwd64h_rows_k)
static int bwd64_half_mode() {
  static const int v = [] {
    const char *e = getenv("GSAGE_BWD64");
    return (e && e[0] == 'f') ? 0 : (e && e[0] == 'd') ? 1 : 2;
  }();
  return v;
}
static int launch_wide(int ncols, const GemmArgs &a, void *pack_base, int64_t Fi, int64_t Fo,
                       const unsigned *amax_in, hipStream_t st) {
  uint16_t *planes = (uint16_t *)((char *)pack_base + pack_f32_bytes(Fi, Fo));
  // the split kernels win where the K loop is long (256 -> 256 at 10 M rows: 14.7 vs 22.8 ms;
  // 128 -> 128: 5.3 vs 6.7 ms) and tie on the shortest one in use (the dX GEMM of a 128-wide
  // layer, K = 128), so every wide GEMM takes them unless GSAGE_WIDE_GEMM=f32 says otherwise
  const int mode = wide_gemm_mode();
  // fp16 kernel (two planes, three products): forwards onto 256 columns with K >= 256, where it
  // beats the bf16 scheme also when it has to find max |A| itself (256 -> 256 at 2.56 M rows:
  // 3.6 ms with its two absmax passes, 2.8 ms with a bound handed in, against 4.6 ms); on
  // 128-column tiles the two tie.  Not for the dX GEMM: rows of a gradient differ by many orders
  // of magnitude, most tiles would be recomputed.
  if ((mode == 2 || (mode == -1 && ncols == 256 && a.K >= 256 && a.epi.mode != MODE_DX)) &&
      half_shape_ok(a, ncols == 128 ? 128 : 256)) {
    HalfCtl *ctl = (HalfCtl *)((char *)planes + pack_planes_bytes(Fi, Fo));
    return ncols == 128 ? launch_gemm_half<2>(a, planes, ctl, amax_in, st)
                        : launch_gemm_half<4>(a, planes, ctl, amax_in, st);
  }
  if (mode != 0 && (mode == 1 || a.K >= 128))
    return ncols == 128 ? launch_gemm_split<2>(a, planes, st) : launch_gemm_split<4>(a, planes, st);
  return ncols == 128 ? launch_gemm_wide<1>(a, st) : launch_gemm_wide<2>(a, st);
}

extern "C" size_t gsage_project_workspace_bytes(int64_t Fi, int64_t Fo) {
  if (Fi <= 0 || Fo <= 0) return 0;
  return pack_bytes(Fi, Fo);
}

extern "C" size_t gsage_relu_sign_bytes(int64_t N, int64_t Fo) {
  if (N < 0 || Fo <= 0) return 0;
  return (size_t)(((N + 3) / 4) * 4 * ((Fo + 63) / 64) + 1) * sizeof(uint64_t);
}

__global__ void zero_word_k(uint32_t *w) { *w = 0u; }

// forward shapes whose kernel publishes max |out| (EpiArgs::amax_out): the first-layer MFMA kernel
// and the wide tile kernels
extern "C" int gsage_project_fwd_range_supported(int64_t Fi, int64_t Fo, int epilogue) {
  const bool first = Fi == 8 && Fo % 64 == 0 && (Fo == 64 || epilogue != GSAGE_EPI_L2NORM);
  const bool wide = mfma_dim(Fi) && (Fo == 128 || Fo == 256);
  // 64 -> 64: the fp16 two-part kernel, which runs when a bound on |x| is handed in (amax_in)
  const bool sq64 = Fi == 64 && Fo == 64 && proj64_half_mode();
  return first || wide || sq64 ? 1 : 0;
}

static int project_fwd_impl(const float *agg, const float *x, const float *w_l,
                            const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                            int64_t Fo, int epilogue, const uint8_t *keep_mask,
                            uint64_t seed, float p, float *out, float *rnorm, void *wpack,
                            int force_generic, uint64_t *relu_sign, int64_t in_blk, int64_t out_blk,
                            const uint32_t *amax_in, uint32_t *amax_out, gsage_stream_t stream,
                            float *u_rows = nullptr, int *u_written = nullptr) {
  if (u_written) *u_written = 0;
  if (N < 0 || Fi <= 0 || Fo <= 0 || !agg || !x || !w_l || !b_l || !w_r || !out || !wpack)
    return GSAGE_ERR_INVALID_ARG;
  if (epilogue < GSAGE_EPI_NONE || epilogue > GSAGE_EPI_L2NORM) return GSAGE_ERR_INVALID_ARG;
  if (epilogue == GSAGE_EPI_RELU_DROPOUT && !(p >= 0.f && p < 1.f)) return GSAGE_ERR_INVALID_ARG;
  if (Fi > 16384 || Fo > 16384) return GSAGE_ERR_UNSUPPORTED;
  if (in_blk || out_blk) {
    // column-blocked operands: the first-layer MFMA kernel (row-major 8-wide input) and the
    // bf16-split GEMMs only
    const bool first = Fi == 8 && !in_blk && Fo % 64 == 0 && epilogue != GSAGE_EPI_L2NORM;
    const bool wide = mfma_dim(Fi) && (Fo == 128 || Fo == 256) && Fi >= 64 && wide_gemm_mode() != 0 &&
                      (!in_blk || Fi >= 128);
    if (force_generic != 0 || !(first || wide)) return GSAGE_ERR_UNSUPPORTED;
  }
  if (amax_out && amax_in == amax_out) return GSAGE_ERR_INVALID_ARG;
  hipStream_t st = gs_stream(stream);
  if (amax_out) {
    // only the kernels behind gsage_project_fwd_range_supported publish it
    if (force_generic != 0 || !gsage_project_fwd_range_supported(Fi, Fo, epilogue)) return GSAGE_ERR_UNSUPPORTED;
    if (Fi == 64 && Fo == 64 && !amax_in) return GSAGE_ERR_UNSUPPORTED;   // (only its fp16 kernel publishes)
    hipLaunchKernelGGL(zero_word_k, dim3(1), dim3(1), 0, st, amax_out);
    GSAGE_CHECK_LAUNCH();
  }
  if (N == 0) return GSAGE_OK;
  // only the first-layer MFMA kernel leaves sign bits; the word behind them tells the backward
  bool sign_written = false;
  uint64_t *sign_flag = relu_sign ? relu_sign + ((N + 3) / 4) * 4 * ((Fo + 63) / 64) : nullptr;
  if (sign_flag) {
    hipLaunchKernelGGL(sign_flag_k, dim3(1), dim3(1), 0, st, sign_flag, (uint64_t)0);
    GSAGE_CHECK_LAUNCH();
  }
  EpiArgs e{};
  e.mode = epilogue;
  e.bias = b_l;
  e.keep_mask = keep_mask;
  e.seed = seed;
  e.p = p;
  e.inv_keep = epilogue == GSAGE_EPI_RELU_DROPOUT ? 1.0f / (1.0f - p) : 1.0f;
  e.rnorm = rnorm;
  e.row_scale = nullptr;
  e.amax_out = amax_out;
  float *wp = (float *)wpack;
  const int fi = (int)Fi, fo = (int)Fo;
  const int pack_grid = gs_grid(2 * Fi * Fo, 256, 64);
  const bool aligned = (((uintptr_t)agg | (uintptr_t)x | (uintptr_t)out | (uintptr_t)b_l |
                         (uintptr_t)keep_mask) & 15) == 0;
  if (force_generic == 0 && aligned && Fi == 64 && Fo == 64 && !in_blk && !out_blk) {
    int64_t ntiles = (N + 63) / 64;
    int grid = (int)(ntiles < S64_GRID ? ntiles : S64_GRID);
    if (amax_in && proj64_half_mode()) {
      // a bound on |x| came with the activations: the fp16 two-part kernel (the planes and the two
      // scale factors live where the wide kernels keep theirs, behind the fp32 pack)
      uint16_t *planes = (uint16_t *)((char *)wpack + pack_f32_bytes(Fi, Fo));
      float *params = (float *)((char *)planes + pack_planes_bytes(Fi, Fo));
      hipLaunchKernelGGL(p64h_pack_k, dim3(1), dim3(256), 0, st, w_l, w_r, amax_in, planes, params);
      hipLaunchKernelGGL(proj64h_fwd_k, dim3(grid), dim3(256), 0, st, agg, x, (const uint16_t *)planes,
                         (const float *)params, N, e, out);
      GSAGE_CHECK_LAUNCH();
      return GSAGE_OK;
    }
    if (amax_out) return GSAGE_ERR_UNSUPPORTED;       // the fp32 kernel does not publish max |out|
    hipLaunchKernelGGL(pack_fwd_k, dim3(pack_grid), dim3(256), 0, st, w_l, w_r, fi, fo, wp);
    hipLaunchKernelGGL(proj64_fwd_k, dim3(grid), dim3(256), 0, st, agg, x, wp, N, e, out);
    GSAGE_CHECK_LAUNCH();
    return GSAGE_OK;
  }
  if (force_generic != 1 && aligned && mfma_dim(Fi) && mfma_dim(Fo)) {
    hipLaunchKernelGGL(pack_fwd_k, dim3(pack_grid), dim3(256), 0, st, w_l, w_r, fi, fo, wp);
    GemmArgs a{};
    a.A1 = agg; a.A2 = x; a.Bp = wp; a.C1 = out; a.C2 = nullptr;
    a.M = N; a.K1 = fi; a.K = 2 * fi; a.N1 = fo; a.Ncols = fo; a.epi = e;
    a.a_blk = in_blk; a.c_blk = out_blk;
    if (force_generic == 0 && (fo == 128 || fo == 256)) return launch_wide(fo, a, wpack, Fi, Fo, amax_in, st);
    if (amax_out) return GSAGE_ERR_UNSUPPORTED;     // gemm_rows_k does not publish max |out|
    return launch_gemm_bn(fo, a, st);
  }
  hipLaunchKernelGGL(pack_t_k, dim3(pack_grid), dim3(256), 0, st, w_l, w_r, fi, fo, wp);
  int grid = gs_grid(N, 4, 256 * 16);
  if (force_generic == 0 && Fi == 8 && Fo % 64 == 0 && (Fo == 64 || epilogue != GSAGE_EPI_L2NORM) &&
      ((((uintptr_t)agg | (uintptr_t)x | (uintptr_t)out | (uintptr_t)b_l | (uintptr_t)keep_mask) & 15) == 0)) {
    int g16 = gs_grid(N, 64, 256 * 8);
    const bool relu = epilogue == GSAGE_EPI_RELU || epilogue == GSAGE_EPI_RELU_DROPOUT;
    // the rows for the next layer's recompute gather, if wanted: its shapes and epilogues only
    float *U = (u_rows && relu && (fo == 64 || fo == 128 || fo == 256) && (((uintptr_t)u_rows) & 15) == 0)
                   ? u_rows : nullptr;
    const int urow = (epilogue == GSAGE_EPI_RELU_DROPOUT && fo > 64) ? 32 : 16;
    if (U) {     // the word behind the table: max |u|, raised by the kernel's waves
      hipLaunchKernelGGL(zero_word_k, dim3(1), dim3(1), 0, st, (uint32_t *)(U + N * urow));
      GSAGE_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(project_fwd_k16_k, dim3(g16, fo / 64), dim3(256), 0, st, agg, x, wp, N, fo, e,
                       out, relu ? relu_sign : nullptr, out_blk, U, urow);
    sign_written = relu;
    if (U && u_written) *u_written = 1;
  } else if (amax_out) {
    return GSAGE_ERR_UNSUPPORTED;     // the kernels below do not publish max |out| (unaligned operands)
  } else if (2 * Fi <= 16 && (Fo <= 64 || (Fo % 64 == 0 && epilogue != GSAGE_EPI_L2NORM)))
    hipLaunchKernelGGL((project_fwd_smallk_k<16>), dim3(grid, (fo + 63) / 64), dim3(256), 0, st, agg,
                       x, wp, N, fi, fo, e, out);
  else
    hipLaunchKernelGGL(project_fwd_generic_k, dim3(grid), dim3(256), 0, st, agg, x, wp, N, fi, fo,
                       e, out);
  GSAGE_CHECK_LAUNCH();
  if (sign_flag && sign_written) {
    hipLaunchKernelGGL(sign_flag_k, dim3(1), dim3(1), 0, st, sign_flag, (uint64_t)1);
    GSAGE_CHECK_LAUNCH();
  }
  return GSAGE_OK;
}

extern "C" int gsage_project_fwd_ex(const float *agg, const float *x, const float *w_l,
                                    const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                                    int64_t Fo, int epilogue, const uint8_t *keep_mask,
                                    uint64_t seed, float p, float *out, float *rnorm, void *wpack,
                                    int force_generic, uint64_t *relu_sign,
                                    gsage_stream_t stream) {
  return project_fwd_impl(agg, x, w_l, b_l, w_r, N, Fi, Fo, epilogue, keep_mask, seed, p, out, rnorm,
                          wpack, force_generic, relu_sign, 0, 0, nullptr, nullptr, stream);
}

extern "C" int gsage_project_fwd_u(const float *agg, const float *x, const float *w_l,
                                   const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                                   int64_t Fo, int epilogue, const uint8_t *keep_mask,
                                   uint64_t seed, float p, float *out, float *rnorm, void *wpack,
                                   int force_generic, uint64_t *relu_sign, float *u_rows,
                                   int *u_written, const uint32_t *amax_in, uint32_t *amax_out,
                                   gsage_stream_t stream) {
  return project_fwd_impl(agg, x, w_l, b_l, w_r, N, Fi, Fo, epilogue, keep_mask, seed, p, out, rnorm,
                          wpack, force_generic, relu_sign, 0, 0, amax_in, amax_out, stream, u_rows,
                          u_written);
}

extern "C" int gsage_project_fwd_blocked_supported(int64_t Fi, int64_t Fo, int in_blocked,
                                                   int out_blocked, int epilogue) {
  if (!in_blocked && !out_blocked) return 1;
  if (epilogue == GSAGE_EPI_RELU_DROPOUT) return 0;
  const bool first = Fi == 8 && !in_blocked && Fo % 64 == 0 && epilogue != GSAGE_EPI_L2NORM;
  const bool wide = mfma_dim(Fi) && (Fo == 128 || Fo == 256) && wide_gemm_mode() != 0 &&
                    (!in_blocked || Fi >= 128);
  return first || wide ? 1 : 0;
}

extern "C" int gsage_project_fwd_range(const float *agg, const float *x, const float *w_l,
                                       const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                                       int64_t Fo, int epilogue, float *out, float *rnorm,
                                       void *wpack, int in_blocked, int out_blocked,
                                       const uint32_t *amax_in, uint32_t *amax_out,
                                       gsage_stream_t stream) {
  if (epilogue == GSAGE_EPI_RELU_DROPOUT) return GSAGE_ERR_UNSUPPORTED;   // inference layouts
  if (!gsage_project_fwd_blocked_supported(Fi, Fo, in_blocked, out_blocked, epilogue))
    return GSAGE_ERR_UNSUPPORTED;
  if ((((uintptr_t)agg | (uintptr_t)x | (uintptr_t)out | (uintptr_t)b_l) & 15) != 0)
    return GSAGE_ERR_UNSUPPORTED;       // the tile kernels need 16-byte aligned operands
  return project_fwd_impl(agg, x, w_l, b_l, w_r, N, Fi, Fo, epilogue, nullptr, 0, 0.f, out, rnorm,
                          wpack, 0, nullptr, in_blocked ? N * 64 : 0, out_blocked ? N * 64 : 0,
                          amax_in, amax_out, stream);
}

extern "C" int gsage_project_fwd_blocked(const float *agg, const float *x, const float *w_l,
                                         const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                                         int64_t Fo, int epilogue, float *out, float *rnorm,
                                         void *wpack, int in_blocked, int out_blocked,
                                         gsage_stream_t stream) {
  return gsage_project_fwd_range(agg, x, w_l, b_l, w_r, N, Fi, Fo, epilogue, out, rnorm, wpack,
                                 in_blocked, out_blocked, nullptr, nullptr, stream);
}

extern "C" int gsage_project_fwd(const float *agg, const float *x, const float *w_l,
                                 const float *b_l, const float *w_r, int64_t N, int64_t Fi,
                                 int64_t Fo, int epilogue, const uint8_t *keep_mask,
                                 uint64_t seed, float p, float *out, float *rnorm, void *wpack,
                                 int force_generic, gsage_stream_t stream) {
  return gsage_project_fwd_ex(agg, x, w_l, b_l, w_r, N, Fi, Fo, epilogue, keep_mask, seed, p, out,
                              rnorm, wpack, force_generic, nullptr, stream);
}

extern "C" size_t gsage_project_bwd_workspace_bytes(int64_t N, int64_t Fi, int64_t Fo) {
  if (N < 0 || Fi <= 0 || Fo <= 0) return 0;
  size_t pack = pack_bytes(Fi, Fo);
  size_t slabs = gs_align_up((size_t)num_slabs(N) * (size_t)(2 * Fi * Fo + Fo) * sizeof(float), 256);
  size_t slabs16 = (Fo % 64 == 0 && Fi <= 8) ? (size_t)DW16_BLOCKS * DW16_SLAB * sizeof(float) : 0;
  return pack + (slabs > slabs16 ? slabs : slabs16);
}

static __global__ void __launch_bounds__(256) fill_words_k(uint32_t *p, int64_t words, uint32_t v) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) p[i] = v;
}

extern "C" int gsage_project_bwd_rows_supported(int64_t Fi, int64_t Fo) {
  return Fi == 64 && Fo == 64 ? 1 : 0;
}

static int project_bwd_impl(const float *grad_out, const float *out, const float *rnorm,
                            const float *agg, const float *x, const float *w_l,
                            const float *w_r, const float *agg_scale, int64_t N, int64_t Fi,
                            int64_t Fo, int epilogue, float p, float *dpre, float *grad_agg,
                            float *grad_x, float *grad_w_l, float *grad_b, float *grad_w_r,
                            void *ws, size_t ws_bytes, int force_generic,
                            uint32_t *nz_mask, const uint64_t *relu_sign,
                            const int32_t *live_rows, const int32_t *n_live, int64_t live_cap,
                            gsage_stream_t stream, uint32_t *gmax = nullptr) {
  if (N < 0 || Fi <= 0 || Fo <= 0 || !grad_out || !agg || !x || !w_l || !w_r || !grad_w_l ||
      !grad_b || !grad_w_r || !ws)
    return GSAGE_ERR_INVALID_ARG;
  if (epilogue < GSAGE_EPI_NONE || epilogue > GSAGE_EPI_L2NORM) return GSAGE_ERR_INVALID_ARG;
  if (epilogue != GSAGE_EPI_NONE && (!out || !dpre)) return GSAGE_ERR_INVALID_ARG;
  if (epilogue == GSAGE_EPI_L2NORM && !rnorm) return GSAGE_ERR_INVALID_ARG;
  if (Fi > 16384 || Fo > 16384) return GSAGE_ERR_UNSUPPORTED;
  if (ws_bytes < gsage_project_bwd_workspace_bytes(N, Fi, Fo)) return GSAGE_ERR_WORKSPACE;
  hipStream_t st = gs_stream(stream);
  const int fi = (int)Fi, fo = (int)Fo;
  float *wp = (float *)ws;
  float *slabs = (float *)((char *)ws + pack_bytes(Fi, Fo));
  const int nslab = num_slabs(N);

  if (gmax) {
    hipLaunchKernelGGL(zero_word_k, dim3(1), dim3(1), 0, st, gmax);
    GSAGE_CHECK_LAUNCH();
  }
  if (N == 0) {
    if (!gs_zero_async(grad_w_l, (size_t)Fi * Fo * 4, st) ||
        !gs_zero_async(grad_w_r, (size_t)Fi * Fo * 4, st) ||
        !gs_zero_async(grad_b, (size_t)Fo * 4, st))
      return GSAGE_ERR_LAUNCH;
    return GSAGE_OK;
  }

  // 0. the 64 -> 64 layer: one fused pass (dpre never leaves the chip)
  {
    const bool al = (((uintptr_t)grad_out | (uintptr_t)out | (uintptr_t)agg | (uintptr_t)x |
                      (uintptr_t)grad_agg | (uintptr_t)grad_x) & 15) == 0;
    if (force_generic == 0 && al && Fi == 64 && Fo == 64 && grad_agg && grad_x) {
      hipLaunchKernelGGL(pack_dx_k, dim3(gs_grid(2 * Fi * Fo, 256, 64)), dim3(256), 0, st, w_l, w_r,
                         fi, fo, wp);
      int grid = nslab < S64_BWD_GRID ? nslab : S64_BWD_GRID;
      const float ik = epilogue == GSAGE_EPI_RELU_DROPOUT ? 1.0f / (1.0f - p) : 1.0f;
      auto kern = live_rows
                      ? (epilogue == GSAGE_EPI_NONE ? bwd64_fused_k<0, true>
                         : epilogue == GSAGE_EPI_L2NORM ? bwd64_fused_k<2, true> : bwd64_fused_k<1, true>)
                      : (epilogue == GSAGE_EPI_NONE ? bwd64_fused_k<0, false>
                         : epilogue == GSAGE_EPI_L2NORM ? bwd64_fused_k<2, false> : bwd64_fused_k<1, false>);
      const uint16_t *hplanes = nullptr;
      const float *hparams = nullptr;
      if (live_rows) {
        // tiles are cut from the list: no more workgroups than the list can fill
        const int64_t lt = (live_cap + 31) / 32;
        if (lt < grid) grid = (int)(lt > 0 ? lt : 1);
        if (bwd64_half_mode()) {
          // dX on the fp16 pipe: the weights' two planes and their scale, behind the fp32 pack
          uint16_t *pl = (uint16_t *)((char *)ws + pack_f32_bytes(Fi, Fo));
          float *pr = (float *)((char *)pl + pack_planes_bytes(Fi, Fo));
          hipLaunchKernelGGL(bwd64h_pack_k, dim3(1), dim3(256), 0, st, w_l, w_r, pl, pr);
          hplanes = pl;
          hparams = pr;
          kern = epilogue == GSAGE_EPI_NONE ? bwd64_fused_k<0, true, true>
                 : epilogue == GSAGE_EPI_L2NORM ? bwd64_fused_k<2, true, true> : bwd64_fused_k<1, true, true>;
        }
      }
      if (live_rows && bwd64_half_mode() == 2) {
        auto kh = epilogue == GSAGE_EPI_NONE ? bwd64h_rows_k<0>
                  : epilogue == GSAGE_EPI_L2NORM ? bwd64h_rows_k<2> : bwd64h_rows_k<1>;
        hipLaunchKernelGGL(kh, dim3(grid), dim3(256), 0, st, grad_out, out, rnorm, agg, x, agg_scale, ik,
                           grad_agg, grad_x, slabs, live_rows, n_live, live_cap, gmax, hplanes, hparams);
      } else {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, grad_out, out, rnorm, agg, x,
                           wp, agg_scale, N, ik, grad_agg, grad_x, slabs, nz_mask, live_rows, n_live,
                           live_cap, gmax, hplanes, hparams);
      }
      GSAGE_CHECK_LAUNCH();
      hipLaunchKernelGGL(reduce_slabs_k, dim3((unsigned)((2 * Fi * Fo + Fo + 63) / 64)), dim3(256), 0,
                         st, slabs, grid, fi, fo, grad_w_l, grad_w_r, grad_b);
      GSAGE_CHECK_LAUNCH();
      return GSAGE_OK;
    }
  }

  // the other kernels do not look for zero rows: every row is reported as possibly non-zero
  if (nz_mask) {
    const int64_t words = (N + 31) / 32;
    hipLaunchKernelGGL(fill_words_k, dim3(gs_grid(words, 1024)), dim3(256), 0, st, nz_mask, words,
                       0xFFFFFFFFu);
    GSAGE_CHECK_LAUNCH();
  }

  // 1. dpre
  const float *dp = grad_out;
  const float inv_keep = epilogue == GSAGE_EPI_RELU_DROPOUT ? 1.0f / (1.0f - p) : 1.0f;
  const bool smallk = 2 * Fi <= 16 && (Fo <= 64 || Fo % 64 == 0);
  const bool relu_epi = epilogue == GSAGE_EPI_RELU || epilogue == GSAGE_EPI_RELU_DROPOUT;
  // first layer in training: no input gradient wanted -> the ReLU derivative is applied
  // inside the dW kernel and dpre is never materialised
  const bool fuse_relu = smallk && relu_epi && !grad_agg && !grad_x;
  if (epilogue != GSAGE_EPI_NONE && !fuse_relu) {
    hipLaunchKernelGGL(epilogue_bwd_k, dim3(gs_grid(N, 4, 256 * 16)), dim3(256), 0, st, grad_out,
                       out, rnorm, N, fo, epilogue, inv_keep, dpre);
    GSAGE_CHECK_LAUNCH();
    dp = dpre;
  }
  const bool aligned = (((uintptr_t)dp | (uintptr_t)agg | (uintptr_t)x | (uintptr_t)grad_agg |
                         (uintptr_t)grad_x) & 15) == 0;
  const bool mfma = force_generic != 1 && aligned && mfma_dim(Fi) && mfma_dim(Fo);

  // 2. input gradients
  if (grad_agg || grad_x) {
    if (mfma && grad_agg && grad_x) {
      hipLaunchKernelGGL(pack_dx_k, dim3(gs_grid(2 * Fi * Fo, 256, 64)), dim3(256), 0, st, w_l, w_r,
                         fi, fo, wp);
      GemmArgs a{};
      a.A1 = dp; a.A2 = nullptr; a.Bp = wp; a.C1 = grad_agg; a.C2 = grad_x;
      a.M = N; a.K1 = fo; a.K = fo; a.N1 = fi; a.Ncols = 2 * fi;
      a.epi.mode = MODE_DX;
      a.epi.row_scale = agg_scale;
      int rc = force_generic != 0 ? launch_gemm_bn(fi, a, st)
                                  : launch_wide(2 * fi, a, ws, Fi, Fo, nullptr, st);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(dx_generic_k, dim3(gs_grid(N, 4, 256 * 16)), dim3(256), 0, st, dp, w_l,
                         w_r, agg_scale, N, fi, fo, grad_agg, grad_x);
      GSAGE_CHECK_LAUNCH();
    }
  }

  // 3. weight gradients: partial slabs, then an ordered sum
  if (force_generic == 0 && Fo % 64 == 0 && Fi <= 8 &&
      ((((uintptr_t)dp | (uintptr_t)out) & 15) == 0)) {
    int ncb = fo / 64;
    int grid = gs_grid(N, 64, DW16_BLOCKS / ncb);
    hipLaunchKernelGGL(dw16_mfma_k, dim3(grid, ncb), dim3(256), 0, st, dp, out, fuse_relu ? 1 : 0,
                       inv_keep, agg, x, N, fi, fo, slabs, fuse_relu ? relu_sign : nullptr);
    GSAGE_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_dw16_k, dim3((DW16_SLAB + 63) / 64, ncb), dim3(256), 0, st, slabs,
                       grid, fi, grad_w_l, grad_w_r, grad_b);
    GSAGE_CHECK_LAUNCH();
    return GSAGE_OK;
  }
  int64_t rpb = (N + nslab - 1) / nslab;
  rpb = (rpb + DW_RK - 1) / DW_RK * DW_RK;
  if (mfma) {
    dim3 grid((unsigned)nslab, (unsigned)((fo / 64) * (2 * fi / 128)));
    hipLaunchKernelGGL(dw_mfma_k, grid, dim3(256), 0, st, dp, agg, x, N, fi, fo, rpb, slabs);
  } else if (2 * Fi <= 16 && Fo <= 64) {
    hipLaunchKernelGGL((dw_smallk_k<16>), dim3(nslab), dim3(256), 0, st, dp, out, fuse_relu ? 1 : 0,
                       inv_keep, agg, x, N, fi, fo, rpb, slabs);
  } else {
    hipLaunchKernelGGL(dw_generic_k, dim3(nslab), dim3(256), 0, st, dp, agg, x, N, fi, fo, rpb,
                       slabs);
  }
  GSAGE_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_slabs_k, dim3((unsigned)((2 * Fi * Fo + Fo + 63) / 64)), dim3(256), 0, st,
                     slabs, nslab, fi, fo, grad_w_l, grad_w_r, grad_b);
  GSAGE_CHECK_LAUNCH();
  return GSAGE_OK;
}

extern "C" int gsage_project_bwd_ex(const float *grad_out, const float *out, const float *rnorm,
                                    const float *agg, const float *x, const float *w_l,
                                    const float *w_r, const float *agg_scale, int64_t N, int64_t Fi,
                                    int64_t Fo, int epilogue, float p, float *dpre, float *grad_agg,
                                    float *grad_x, float *grad_w_l, float *grad_b, float *grad_w_r,
                                    void *ws, size_t ws_bytes, int force_generic,
                                    uint32_t *nz_mask, const uint64_t *relu_sign,
                                    gsage_stream_t stream) {
  return project_bwd_impl(grad_out, out, rnorm, agg, x, w_l, w_r, agg_scale, N, Fi, Fo, epilogue, p,
                          dpre, grad_agg, grad_x, grad_w_l, grad_b, grad_w_r, ws, ws_bytes,
                          force_generic, nz_mask, relu_sign, nullptr, nullptr, 0, stream);
}

extern "C" int gsage_project_bwd_rows(const float *grad_out, const float *out, const float *rnorm,
                                      const float *agg, const float *x, const float *w_l,
                                      const float *w_r, const float *agg_scale, int64_t N,
                                      int64_t Fi, int64_t Fo, int epilogue, float p, float *grad_agg,
                                      float *grad_x, float *grad_w_l, float *grad_b,
                                      float *grad_w_r, void *ws, size_t ws_bytes,
                                      const int32_t *live_rows, const int32_t *n_live,
                                      int64_t live_cap, gsage_stream_t stream) {
  if (!gsage_project_bwd_rows_supported(Fi, Fo)) return GSAGE_ERR_UNSUPPORTED;
  if (!live_rows || !n_live || live_cap <= 0 || !grad_agg || !grad_x) return GSAGE_ERR_INVALID_ARG;
  if ((((uintptr_t)grad_out | (uintptr_t)out | (uintptr_t)agg | (uintptr_t)x |
        (uintptr_t)grad_agg | (uintptr_t)grad_x) & 15) != 0)
    return GSAGE_ERR_UNSUPPORTED;
  // dpre never leaves the chip in the fused kernel: any non-null pointer satisfies the check
  float *dpre_unused = epilogue != GSAGE_EPI_NONE ? grad_agg : nullptr;
  return project_bwd_impl(grad_out, out, rnorm, agg, x, w_l, w_r, agg_scale, N, Fi, Fo, epilogue, p,
                          dpre_unused, grad_agg, grad_x, grad_w_l, grad_b, grad_w_r, ws, ws_bytes, 0,
                          nullptr, nullptr, live_rows, n_live, live_cap, stream);
}

extern "C" int gsage_project_bwd_rows_range(const float *grad_out, const float *out, const float *rnorm,
                                            const float *agg, const float *x, const float *w_l,
                                            const float *w_r, const float *agg_scale, int64_t N,
                                            int64_t Fi, int64_t Fo, int epilogue, float p,
                                            float *grad_agg, float *grad_x, float *grad_w_l,
                                            float *grad_b, float *grad_w_r, void *ws, size_t ws_bytes,
                                            const int32_t *live_rows, const int32_t *n_live,
                                            int64_t live_cap, uint32_t *gmax_out,
                                            gsage_stream_t stream) {
  if (!gsage_project_bwd_rows_supported(Fi, Fo)) return GSAGE_ERR_UNSUPPORTED;
  if (!live_rows || !n_live || live_cap <= 0 || !grad_agg || !grad_x || !gmax_out) return GSAGE_ERR_INVALID_ARG;
  if ((((uintptr_t)grad_out | (uintptr_t)out | (uintptr_t)agg | (uintptr_t)x |
        (uintptr_t)grad_agg | (uintptr_t)grad_x) & 15) != 0)
    return GSAGE_ERR_UNSUPPORTED;
  float *dpre_unused = epilogue != GSAGE_EPI_NONE ? grad_agg : nullptr;
  return project_bwd_impl(grad_out, out, rnorm, agg, x, w_l, w_r, agg_scale, N, Fi, Fo, epilogue, p,
                          dpre_unused, grad_agg, grad_x, grad_w_l, grad_b, grad_w_r, ws, ws_bytes, 0,
                          nullptr, nullptr, live_rows, n_live, live_cap, stream, gmax_out);
}

extern "C" int gsage_project_bwd(const float *grad_out, const float *out, const float *rnorm,
                                 const float *agg, const float *x, const float *w_l,
                                 const float *w_r, const float *agg_scale, int64_t N, int64_t Fi,
                                 int64_t Fo, int epilogue, float p, float *dpre, float *grad_agg,
                                 float *grad_x, float *grad_w_l, float *grad_b, float *grad_w_r,
                                 void *ws, size_t ws_bytes, int force_generic,
                                 gsage_stream_t stream) {
  return gsage_project_bwd_ex(grad_out, out, rnorm, agg, x, w_l, w_r, agg_scale, N, Fi, Fo, epilogue,
                              p, dpre, grad_agg, grad_x, grad_w_l, grad_b, grad_w_r, ws, ws_bytes,
                              force_generic, nullptr, nullptr, stream);
}

#ifdef HB_PROF
extern "C" int gsage_debug_hb_prof(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(hb_prof), sizeof(unsigned long long) * 2 * 256 * 8) == hipSuccess ? 0 : -1;
}
#endif
