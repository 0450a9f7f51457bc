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
  int *flag;
};

__device__ __forceinline__ void sg_block_reduce(float (&st)[SG_NSTAT], float *partials) {
  __shared__ float red[4][SG_NSTAT];
  const int lane = gs_lane(), w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < SG_NSTAT; ++i) {
    float v = gs_wave_sum(st[i]);
    if (lane == 0) red[w][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < SG_NSTAT)
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
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        dotv[u] = s;
      }
      if (sub == 0) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          int64_t r = r0 + u;
          if (ok[u]) sg_terms(a, b, (int)r, dotv[u], st);
          else if (active && r < R) a.coef[b * R + r] = 0.f;
        }
      }
    }
  }
  if (bad) atomicOr(a.flag, GSAGE_FLAG_INDEX_OUT_OF_RANGE);
  sg_block_reduce(st, a.partials);
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
  if (bad) atomicOr(a.flag, GSAGE_FLAG_INDEX_OUT_OF_RANGE);
  sg_block_reduce(st, a.partials);
}

// one wavefront: ordered double-precision sum of the per-workgroup partials
__global__ void __launch_bounds__(64)
skipgram_finish_k(const float *__restrict__ partials, int nblocks, float invB, float invBK,
                  const int *__restrict__ flag, float *__restrict__ stats) {
  const int lane = gs_lane();
  double s[SG_NSTAT];
#pragma unroll
  for (int i = 0; i < SG_NSTAT; ++i) {
    double v = 0.0;
    for (int b = lane; b < nblocks; b += 64) v += (double)partials[b * 8 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    s[i] = v;
  }
  if (lane == 0) {
    float pos_loss = (float)(s[0] * (double)invB);
    float neg_loss = (float)(s[1] * (double)invBK);
    stats[0] = pos_loss + neg_loss;
    stats[1] = pos_loss;
    stats[2] = neg_loss;
    stats[3] = (float)(s[2] * (double)invB);
    stats[4] = (float)(s[3] * (double)invBK);
    stats[5] = (float)(s[4] * (double)invB);
    stats[6] = (float)(s[5] * (double)invBK);
    stats[7] = (float)(*flag);
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

int sg_blocks(int64_t B, int pairs_per_block) {
  int64_t nb = (B + pairs_per_block - 1) / pairs_per_block;
  if (nb < 1) nb = 1;
  if (nb > SG_MAX_BLOCKS) nb = SG_MAX_BLOCKS;
  return (int)nb;
}

}  // namespace

extern "C" size_t gsage_skipgram_workspace_bytes(int64_t B) {
  (void)B;
  return (size_t)SG_MAX_BLOCKS * 8 * sizeof(float) + 256;
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
  int *flag = (int *)((char *)ws + (size_t)SG_MAX_BLOCKS * 8 * sizeof(float));
  if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return GSAGE_ERR_LAUNCH;
  SgArgs a;
  a.emb = emb; a.N = N; a.D = D; a.tg = targets; a.ctx = contexts; a.neg = negatives;
  a.B = B; a.K = K; a.scale = scale;
  a.invB = (float)(1.0 / loss_norm_B);
  a.invBK = K > 0 ? (float)(1.0 / loss_norm_BK) : 0.f;
  a.coef = coef; a.partials = partials; a.flag = flag;
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
  hipLaunchKernelGGL(skipgram_finish_k, dim3(1), dim3(64), 0, st, partials, nb, a.invB, a.invBK,
                     flag, stats);
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
