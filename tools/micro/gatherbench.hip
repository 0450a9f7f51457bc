// Does the memory type of a table change what a random gather of SHORT rows costs?  Every fabric
// read of a normally cached allocation is a 128-byte line (profiles/r02_c4_pmc_hbm.json), so a
// 32-byte row costs four times its bytes.  Here: R random rows of 32 / 64 bytes out of a table of
// N rows, the table allocated (0) with hipMalloc, (1) uncached, (2) fine-grained.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LPR>   // lanes per row, 16 bytes each
__global__ void __launch_bounds__(256) gather_k(const int *__restrict__ idx, long R, const float4 *__restrict__ tab,
                                                float *__restrict__ out) {
  const int lane = threadIdx.x & 63, sub = lane % LPR, slot = lane / LPR;
  constexpr int G = 64 / LPR;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long base = w * (long)(G * 8);            // 8 row groups per wave, 4 + 4 in flight
  float acc = 0.f;
  if (base + G * 8 <= R) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = idx[base + (h * 4 + k) * G + slot];
        r[k] = tab[(long)row * LPR + sub];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += r[k].x + r[k].y + r[k].z + r[k].w;
    }
  }
  if (acc == 12345.678f) out[w] = acc;            // keeps the loads alive, (almost) never stores
}

template <int LPR>
static void run(const char *what, const int *idx, long R, const float4 *tab, float *out) {
  constexpr int G = 64 / LPR;
  const long waves = R / (G * 8);
  const unsigned grid = (unsigned)((waves + 3) / 4);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_k<LPR>), dim3(grid), dim3(256), 0, 0, idx, R, tab, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gather_k<LPR>), dim3(grid), dim3(256), 0, 0, idx, R, tab, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-28s rows of %3d B: %7.3f ms  %6.1f G rows/s  %5.2f TB/s of row bytes\n", what, LPR * 16, ms,
         R / ms / 1e6, (double)R * LPR * 16 / ms / 1e9);
}

int main() {
  const long N = 10000000, R = 200000000;
  int *idx; float *out;
  CK(hipMalloc(&idx, R * 4)); CK(hipMalloc(&out, (R / 16 + 64) * 4));
  {
    std::vector<int> h(R);
    unsigned long long s = 88172645463325252ull;
    for (long i = 0; i < R; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % N); }
    CK(hipMemcpy(idx, h.data(), R * 4, hipMemcpyHostToDevice));
  }
  for (int mode = 0; mode < 3; ++mode) {
    float4 *tab = nullptr;
    const size_t bytes = (size_t)N * 64;
    hipError_t e = mode == 0 ? hipMalloc((void **)&tab, bytes)
                 : hipExtMallocWithFlags((void **)&tab, bytes, mode == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    const char *name = mode == 0 ? "hipMalloc" : mode == 1 ? "hipDeviceMallocUncached" : "hipDeviceMallocFinegrained";
    if (e != hipSuccess) { printf("%s: allocation refused (%s)\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CK(hipMemset(tab, 0, bytes));
    run<2>(name, idx, R, tab, out);
    run<4>(name, idx, R, tab, out);
    CK(hipFree(tab));
  }
  return 0;
}
