// How fast can 256-byte rows be accumulated at random positions of a 10M x 64 fp32 table?
//   a1: one wave per row, lane = column, global_atomic_add_f32 (the loss backward's shape)
//   a2: same rows, plain load + add + store (not safe with duplicates; the upper bound)
//   a3: atomics, but rows visited in ascending order (sorted indices)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const long *idx, long R, float *tab) {
  const int lane = threadIdx.x & 63;
  const long w0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  for (long r = w0; r < R; r += nw) {
    float *p = tab + idx[r] * 64 + lane;
    if (MODE == 0) atomicAdd(p, 1.0f);
    else *p = *p + 1.0f;
  }
}

int main() {
  const long N = 10000000, R = 7340032;   // 2^20 pairs x 7 rows
  float *tab; long *idx;
  CK(hipMalloc(&tab, N * 64 * 4)); CK(hipMemset(tab, 0, N * 64 * 4));
  CK(hipMalloc(&idx, R * 8));
  std::vector<long> h(R);
  unsigned long long s = 88172645463325252ull;
  for (long i = 0; i < R; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (long)(s % N); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-52s %7.3f ms  %6.2f TB/s of row payload\n", name, ms, R * 256.0 / 1e9 / ms);
  };
  CK(hipMemcpy(idx, h.data(), R * 8, hipMemcpyHostToDevice));
  for (int grid : {4096, 16384})  {
    char nm[96];
    snprintf(nm, 96, "a1 atomic rows, random order, %d blocks", grid);
    run(nm, [&] { k<0><<<grid, 256>>>(idx, R, tab); });
    snprintf(nm, 96, "a2 load+add+store rows, random order, %d blocks", grid);
    run(nm, [&] { k<1><<<grid, 256>>>(idx, R, tab); });
  }
  std::sort(h.begin(), h.end());
  CK(hipMemcpy(idx, h.data(), R * 8, hipMemcpyHostToDevice));
  run("a3 atomic rows, ascending order, 4096 blocks", [&] { k<0><<<4096, 256>>>(idx, R, tab); });
  run("a4 load+add+store rows, ascending order, 4096 blocks", [&] { k<1><<<4096, 256>>>(idx, R, tab); });
  return 0;
}
