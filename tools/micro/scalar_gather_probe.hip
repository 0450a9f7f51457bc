// Can the SCALAR memory path (s_load through the scalar data cache, 64-byte lines) fetch random
// short rows beside or instead of the vector path, whose every L1 miss costs a 128-byte line and
// which tops out at ~53 G rows/s chip-wide (line_fetch_probe.hip)?  Read-only probe, no libgsage:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/scalar_gather_probe.hip -o build/scalar_gather_probe
//   build/scalar_gather_probe [rows = 10000000] [gathers = 200000000]
// Modes: vector only (4 lanes x 16 B per 64-byte row, 16 rows per wave-instruction), scalar only (one
// s_load_dwordx16 per row, wave-uniform address), and mixed (SV scalar rows per 16 vector rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define CONST_AS __attribute__((address_space(4)))

// VU vector steps (16 rows each) and SU scalar rows per trip of a wave; 64-byte rows
template <int VU, int SU>
__global__ void __launch_bounds__(256) gather_k(const char *__restrict__ table, const int *__restrict__ idx,
                                                int64_t n_idx, int *__restrict__ out) {
  constexpr int PER = 16 * VU + SU;                     // indices a wave consumes per trip
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * 256) >> 6;
  int acc = 0;
  for (int64_t base = wave * PER; base + PER <= n_idx; base += nwaves * PER) {
    i32x4 v[VU > 0 ? VU : 1];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      const int id = idx[base + 16 * u + (lane >> 2)];
      v[u] = *(const i32x4 *)(table + ((int64_t)id * 64 + 16 * (lane & 3)));
    }
    i32x16 s[SU > 0 ? SU : 1];
    if (SU > 0) {
      const int64_t sb = __builtin_amdgcn_readfirstlane((int)(base + 16 * VU));   // (fits: n_idx < 2^31)
      const CONST_AS int *ip = (const CONST_AS int *)(uintptr_t)(idx + sb);
      int sid[SU > 0 ? SU : 1];
#pragma unroll
      for (int u = 0; u < SU; ++u) sid[u] = ip[u];
#pragma unroll
      for (int u = 0; u < SU; ++u)
        s[u] = *(const CONST_AS i32x16 *)(uintptr_t)(table + (int64_t)sid[u] * 64);
    }
#pragma unroll
    for (int u = 0; u < VU; ++u) acc += v[u][0] + v[u][1] + v[u][2] + v[u][3];
    if (SU > 0) {
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s[u][k];
        acc += t;
      }
    }
  }
  if (acc == 0x12345678) out[0] = acc;
}

template <int VU, int SU>
static void run(const char *name, const char *table, const int *idx, int64_t n_idx, int *out, int waves_per_cu) {
  const int grid = 256 * waves_per_cu / 4;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_k<VU, SU>), dim3(grid), dim3(256), 0, 0, table, idx, n_idx, out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL((gather_k<VU, SU>), dim3(grid), dim3(256), 0, 0, table, idx, n_idx, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 3;
  printf("%-28s waves/CU %2d: %8.3f ms  %6.1f G rows/s  (scalar share %.0f %%)\n", name, waves_per_cu, ms,
         n_idx / ms / 1e6, 100.0 * SU / (16 * VU + SU));
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 10000000;
  const int64_t n_idx = argc > 2 ? atoll(argv[2]) : 200000000;
  char *table;
  int *idx, *out;
  CHECK(hipMalloc((void **)&table, rows * 64));
  CHECK(hipMalloc((void **)&idx, n_idx * 4));
  CHECK(hipMalloc((void **)&out, 64));
  CHECK(hipMemset(table, 1, rows * 64));
  int *h = (int *)malloc(n_idx * 4);
  uint64_t st = 88172645463325252ull;
  for (int64_t i = 0; i < n_idx; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    h[i] = (int)(st % (uint64_t)rows);
  }
  CHECK(hipMemcpy(idx, h, n_idx * 4, hipMemcpyHostToDevice));
  free(h);
  for (int wpc : {16, 32}) {
    run<4, 0>("vector only", table, idx, n_idx, out, wpc);
    run<0, 8>("scalar only (8 in flight)", table, idx, n_idx, out, wpc);
    run<0, 4>("scalar only (4 in flight)", table, idx, n_idx, out, wpc);
    run<4, 4>("mixed 64 vector + 4 scalar", table, idx, n_idx, out, wpc);
    run<4, 8>("mixed 64 vector + 8 scalar", table, idx, n_idx, out, wpc);
    run<2, 8>("mixed 32 vector + 8 scalar", table, idx, n_idx, out, wpc);
  }
  return 0;
}
