// Micro-benchmarks of the stream shapes the projection kernels use (N rows x 64 floats):
//   v1 elementwise 2 reads : 1 write, float4, grid-stride
//   v2 persistent workgroups, 32-row tiles: burst of loads -> LDS -> barrier -> float4 stores
//   v3 like v2, stores as dword with 128-byte row segments (MFMA accumulator layout)
//   v4 grid-stride, 1 narrow read, stores as 64-byte chunks per row (first-layer forward layout)
//   v5 like v2 with 4 input streams and 2 output streams (the fused backward's traffic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256) v1(const float4 *a, const float4 *b, float4 *c, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// v0: read-only (sum kept in a register, one conditional store so the loads stay)
template <int UNROLL>
__global__ void __launch_bounds__(256) v0(const float4 *a, float *out, size_t n4) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    float4 x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += x[u].x + x[u].y + x[u].z + x[u].w;
  }
  for (; i < n4; i += stride) { float4 x = a[i]; acc += x.x + x.y + x.z + x.w; }
  if (acc == 123.456f) out[0] = acc;
}

template <int STORE>
__global__ void __launch_bounds__(256) v2(const float *a, const float *b, float *c, long M) {
  __shared__ __align__(16) float T[32 * 68];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, rsub = t >> 4, c4 = (t & 15) * 4;
  const long ntiles = (M + 31) / 32;
  float4 ra[2], rb[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    long row = (long)blockIdx.x * 32 + p * 16 + rsub;
    ra[p] = *(const float4 *)(a + row * 64 + c4);
    rb[p] = *(const float4 *)(b + row * 64 + c4);
  }
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
      *(float4 *)(T + (p * 16 + rsub) * 68 + c4) =
          make_float4(ra[p].x + rb[p].x, ra[p].y + rb[p].y, ra[p].z + rb[p].z, ra[p].w + rb[p].w);
    __syncthreads();
    long nt = tile + gridDim.x;
    if (nt < ntiles) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        long row = nt * 32 + p * 16 + rsub;
        ra[p] = *(const float4 *)(a + row * 64 + c4);
        rb[p] = *(const float4 *)(b + row * 64 + c4);
      }
    }
    if (STORE == 0) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        int r = p * 16 + rsub;
        *(float4 *)(c + (tile * 32 + r) * 64 + c4) = *(const float4 *)(T + r * 68 + c4);
      }
    } else {
      // wave w: columns (w&1)*32 + (lane&31), rows (w>>1)*16 + ... : 8 dword stores per lane
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int lr = (w >> 1) * 16 + r * 2 + (lane >> 5);
        int col = (w & 1) * 32 + (lane & 31);
        c[(tile * 32 + lr) * 64 + col] = T[lr * 68 + col];
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) v4(const float *a8, float *c, long M) {
  const int lane = threadIdx.x & 63, rl = lane & 15, q = lane >> 4;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long r0 = wave0 * 16; r0 < M; r0 += nw * 16) {
    long row = r0 + rl;
    float4 x = *(const float4 *)(a8 + row * 8 + (q & 1) * 4);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      *(float4 *)(c + row * 64 + 16 * nt + 4 * q) = make_float4(x.x + nt, x.y, x.z, x.w);
  }
}

// v4b: the same tile, transposed through a wave-private LDS patch so that every store
// instruction writes 4 whole 256-byte rows
__global__ void __launch_bounds__(256) v4b(const float *a8, float *c, long M) {
  __shared__ __align__(16) float T[4][16 * 68];
  const int lane = threadIdx.x & 63, rl = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
  const long wave0 = (long)blockIdx.x * 4 + w, nw = (long)gridDim.x * 4;
  float *tw = T[w];
  for (long r0 = wave0 * 16; r0 < M; r0 += nw * 16) {
    long row = r0 + rl;
    float4 x = *(const float4 *)(a8 + row * 8 + (q & 1) * 4);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      *(float4 *)(tw + rl * 68 + 16 * nt + 4 * q) = make_float4(x.x + nt, x.y, x.z, x.w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int rr = 4 * i + (lane >> 4);
      *(float4 *)(c + (r0 + rr) * 64 + (lane & 15) * 4) = *(const float4 *)(tw + rr * 68 + (lane & 15) * 4);
    }
  }
}

template <int CONTIG>
__global__ void __launch_bounds__(256)
v5(const float *a, const float *b, const float *c, const float *d, float *o1, float *o2, long M) {
  __shared__ __align__(16) float T[2][32 * 68];
  const int t = threadIdx.x, rsub = t >> 4, c4 = (t & 15) * 4;
  const long ntiles = (M + 31) / 32;
  float4 r[4][2];
  const float *src[4] = {a, b, c, d};
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      r[s][p] = *(const float4 *)(src[s] + ((CONTIG ? (long)blockIdx.x * ((ntiles + gridDim.x - 1) / gridDim.x) : (long)blockIdx.x) * 32 + p * 16 + rsub) * 64 + c4);
  const long per = (ntiles + gridDim.x - 1) / gridDim.x;
  const long t_begin = CONTIG ? (long)blockIdx.x * per : (long)blockIdx.x;
  const long t_end = CONTIG ? (t_begin + per < ntiles ? t_begin + per : ntiles) : ntiles;
  const long t_step = CONTIG ? 1 : (long)gridDim.x;
  for (long tile = t_begin; tile < t_end; tile += t_step) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *(float4 *)(T[0] + (p * 16 + rsub) * 68 + c4) =
          make_float4(r[0][p].x + r[1][p].x, r[0][p].y + r[1][p].y, r[0][p].z + r[1][p].z, r[0][p].w + r[1][p].w);
      *(float4 *)(T[1] + (p * 16 + rsub) * 68 + c4) =
          make_float4(r[2][p].x + r[3][p].x, r[2][p].y + r[3][p].y, r[2][p].z + r[3][p].z, r[2][p].w + r[3][p].w);
    }
    __syncthreads();
    long nt = tile + t_step;
    if (nt < t_end) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          r[s][p] = *(const float4 *)(src[s] + (nt * 32 + p * 16 + rsub) * 64 + c4);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int rr = p * 16 + rsub;
      *(float4 *)(o1 + (tile * 32 + rr) * 64 + c4) = *(const float4 *)(T[0] + rr * 68 + c4);
      *(float4 *)(o2 + (tile * 32 + rr) * 64 + c4) = *(const float4 *)(T[1] + rr * 68 + c4);
    }
    __syncthreads();
  }
}

int main(int argc, char **argv) {
  const long M = 10000000;   // rows, multiple of 32
  const size_t bytes = (size_t)M * 64 * 4;
  float *buf[6];
  for (int i = 0; i < 6; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], 0, bytes)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, double gb, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, gb / ms);
  };
  const double g1 = bytes / 1e9;
  run("v0 read-only, 4 loads in flight, 2048 blocks", g1, [&] { v0<4><<<2048, 256>>>((float4 *)buf[0], buf[5], (size_t)M * 16); });
  run("v0 read-only, 4 loads in flight, 16384 blocks", g1, [&] { v0<4><<<16384, 256>>>((float4 *)buf[0], buf[5], (size_t)M * 16); });
  run("v0 read-only, 1 load per thread (625000 blocks)", g1, [&] { v0<1><<<625000, 256>>>((float4 *)buf[0], buf[5], (size_t)M * 16); });
  run("v0 read-only, 8 loads in flight, 4096 blocks", g1, [&] { v0<8><<<4096, 256>>>((float4 *)buf[0], buf[5], (size_t)M * 16); });
  run("v1 elementwise 2r:1w grid-stride 16384 blocks", 3 * g1, [&] { v1<<<16384, 256>>>((float4 *)buf[0], (float4 *)buf[1], (float4 *)buf[2], (size_t)M * 16); });
  run("v1 elementwise 2r:1w, 65536 blocks", 3 * g1, [&] { v1<<<65536, 256>>>((float4 *)buf[0], (float4 *)buf[1], (float4 *)buf[2], (size_t)M * 16); });
  run("v1 elementwise 2r:1w, one float4 per thread (625000 blocks)", 3 * g1, [&] { v1<<<625000, 256>>>((float4 *)buf[0], (float4 *)buf[1], (float4 *)buf[2], (size_t)M * 16); });
  run("v1 elementwise 2r:1w, 2048 blocks", 3 * g1, [&] { v1<<<2048, 256>>>((float4 *)buf[0], (float4 *)buf[1], (float4 *)buf[2], (size_t)M * 16); });
  for (int grid : {512, 768, 1024, 2048}) {
    char nm[96];
    snprintf(nm, 96, "v2 tiled float4 stores, %d persistent blocks", grid);
    run(nm, 3 * g1, [&] { v2<0><<<grid, 256>>>(buf[0], buf[1], buf[2], M); });
    snprintf(nm, 96, "v3 tiled dword/128B stores, %d blocks", grid);
    run(nm, 3 * g1, [&] { v2<1><<<grid, 256>>>(buf[0], buf[1], buf[2], M); });
  }
  run("v4 narrow read, 64B-chunk stores, 2048 blocks", 1.125 * g1, [&] { v4<<<2048, 256>>>(buf[0], buf[2], M); });
  run("v4 narrow read, 64B-chunk stores, 16384 blocks", 1.125 * g1, [&] { v4<<<16384, 256>>>(buf[0], buf[2], M); });
  run("v4b narrow read, whole-row stores via LDS, 2048 blocks", 1.125 * g1, [&] { v4b<<<2048, 256>>>(buf[0], buf[2], M); });
  run("v4b narrow read, whole-row stores via LDS, 8192 blocks", 1.125 * g1, [&] { v4b<<<8192, 256>>>(buf[0], buf[2], M); });
  run("v2 tiled float4 stores, one tile per block (312500 blocks)", 3 * g1, [&] { v2<0><<<312500, 256>>>(buf[0], buf[1], buf[2], M); });
  run("v5 4 in / 2 out, one tile per block (312500 blocks)", 6 * g1, [&] { v5<0><<<312500, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], M); });
  run("v5 4 in / 2 out, 16384 blocks", 6 * g1, [&] { v5<0><<<16384, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], M); });
  run("v5c contiguous ranges, 16384 blocks", 6 * g1, [&] { v5<1><<<16384, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], M); });
  for (int grid : {512, 768, 1536})  {
    char nm[96];
    snprintf(nm, 96, "v5 4 in / 2 out tiled, %d blocks", grid);
    run(nm, 6 * g1, [&] { v5<0><<<grid, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], M); });
    snprintf(nm, 96, "v5c same, contiguous tile ranges, %d blocks", grid);
    run(nm, 6 * g1, [&] { v5<1><<<grid, 256>>>(buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], M); });
  }
  return 0;
}
