// What does a random gather of small rows cost the fabric, and can it be made to fetch less than a
// 128-byte line per row?  Standalone probe (no libgsage):
//   hipcc -O3 --offload-arch=gfx950 tools/line_fetch_probe.hip -o build/line_fetch_probe
//   build/line_fetch_probe [rows = 10000000] [gathers = 200000000]
// For row sizes 32 and 64 bytes, allocation kinds (hipMalloc / uncached / fine-grained) and load
// flavours (plain, nontemporal, sc1, sc0 sc1, nt sc0 sc1) it prints the time of one pass of `gathers`
// random row reads and rows/s.  Run it under rocprofv3 --pmc TCC_EA0_RDREQ_sum ... for the request
// sizes (kernel names carry the flavour).
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

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int FLAVOUR> __device__ __forceinline__ f32x4 load16(const char *p) {
  f32x4 v;
  if (FLAVOUR == 0) v = *(const f32x4 *)p;
  else if (FLAVOUR == 1) v = __builtin_nontemporal_load((const f32x4 *)p);
  else if (FLAVOUR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (FLAVOUR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// LPR lanes per row (row bytes / 16); each lane group walks its share of idx, UNR rows in flight
template <int LPR, int FLAVOUR, int UNR>
__global__ void __launch_bounds__(256) gather_k(const char *__restrict__ table, const int *__restrict__ idx,
                                                int64_t n_idx, float *__restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t grp = tid / LPR, ngrp = (int64_t)gridDim.x * 256 / LPR;
  const int sub = (int)(tid % LPR);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = grp * UNR; i + UNR <= n_idx; i += ngrp * UNR) {
    int id[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) id[u] = idx[i + u];
    f32x4 v[UNR];
    if (FLAVOUR <= 1) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) v[u] = load16<FLAVOUR>(table + ((int64_t)id[u] * (16 * LPR) + 16 * sub));
    } else {
      // the asm flavours: issue all loads, one wait (the waitcnt inside load16 would serialise them)
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const char *p = table + ((int64_t)id[u] * (16 * LPR) + 16 * sub);
        if (FLAVOUR == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(p) : "memory");
        else if (FLAVOUR == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v[u]) : "v"(p) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[tid] = acc[0];    // keep the loads alive
}

__global__ void fill_idx_k(int *idx, int64_t n, uint32_t rows, uint64_t seed) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t z = seed + (uint64_t)i * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    idx[i] = (int)(z % rows);
  }
}
__global__ void fill_f_k(float *p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (float)(i & 1023);
}

template <int LPR, int FLAVOUR>
static void run(const char *name, const char *alloc, const char *table, const int *idx, int64_t n_idx, float *out) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_k<LPR, FLAVOUR, 8>), dim3(grid), dim3(256), 0, 0, table, idx, n_idx, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
  }
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-12s rows of %3d B  %-14s %8.3f ms  %6.1f G rows/s\n", alloc, 16 * LPR, name, ms, n_idx / ms * 1e-6);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 10000000;
  const int64_t n_idx = argc > 2 ? atoll(argv[2]) : 200000000;
  int *idx;
  float *out;
  CHECK(hipMalloc(&idx, n_idx * sizeof(int)));
  CHECK(hipMalloc(&out, (size_t)256 * 16 * 256 * sizeof(float)));
  hipLaunchKernelGGL(fill_idx_k, dim3(4096), dim3(256), 0, 0, idx, n_idx, (uint32_t)rows, 42ull);
  CHECK(hipDeviceSynchronize());
  for (int kind = 0; kind < 3; ++kind) {
    const char *alloc = kind == 0 ? "hipMalloc" : kind == 1 ? "uncached" : "finegrained";
    char *table = nullptr;
    const size_t bytes = (size_t)rows * 64;
    hipError_t e = kind == 0   ? hipMalloc(&table, bytes)
                   : kind == 1 ? hipExtMallocWithFlags((void **)&table, bytes, hipDeviceMallocUncached)
                               : hipExtMallocWithFlags((void **)&table, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      printf("%-12s allocation failed: %s\n", alloc, hipGetErrorString(e));
      (void)hipGetLastError();
      continue;
    }
    hipLaunchKernelGGL(fill_f_k, dim3(4096), dim3(256), 0, 0, (float *)table, (int64_t)(bytes / 4));
    CHECK(hipDeviceSynchronize());
    run<2, 0>("plain", alloc, table, idx, n_idx, out);
    run<2, 1>("nontemporal", alloc, table, idx, n_idx, out);
    run<2, 2>("sc1", alloc, table, idx, n_idx, out);
    run<2, 3>("sc0 sc1", alloc, table, idx, n_idx, out);
    run<2, 4>("sc0 sc1 nt", alloc, table, idx, n_idx, out);
    run<4, 0>("plain", alloc, table, idx, n_idx, out);
    run<4, 1>("nontemporal", alloc, table, idx, n_idx, out);
    run<4, 2>("sc1", alloc, table, idx, n_idx, out);
    run<4, 3>("sc0 sc1", alloc, table, idx, n_idx, out);
    run<4, 4>("sc0 sc1 nt", alloc, table, idx, n_idx, out);
    CHECK(hipFree(table));
  }
  return 0;
}
