// A host program with no Python and no torch in it: plain HIP runtime calls for memory, and the
// C ABI of libgsage.so (include/gsage.h) for one SAGEConv layer forward + backward and the
// skip-gram loss on the hand-computed known answers KAT-1 and KAT-5 (oracle/kat.py).  This is what
// a C/C++/Go/Rust host would link against; the Python package binds exactly the same symbols.
//
//   hipcc --offload-arch=gfx950 -Iinclude examples/c_abi_demo.cpp -Lgraphsage_amd/lib -lgsage \
//         -Wl,-rpath,'$ORIGIN/../graphsage_amd/lib' -o examples/c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gsage.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define GS_OK(x) do { int rc_ = (x); if (rc_ != 0) { \
  std::printf("gsage error %d (%s) at line %d\n", rc_, gsage_error_string(rc_), __LINE__); return 3; } } while (0)

template <typename T> static T *to_device(const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc(&d, (h.empty() ? 1 : h.size()) * sizeof(T)) != hipSuccess) std::exit(2);
  if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::exit(2);
  return d;
}
template <typename T> static T *device_alloc(size_t n) {
  T *d = nullptr;
  if (hipMalloc(&d, (n ? n : 1) * sizeof(T)) != hipSuccess) std::exit(2);
  if (hipMemset(d, 0, (n ? n : 1) * sizeof(T)) != hipSuccess) std::exit(2);
  return d;
}
template <typename T> static std::vector<T> to_host(const T *d, size_t n) {
  std::vector<T> h(n);
  if (hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::exit(2);
  return h;
}
template <typename T>
static int expect(const char *what, const std::vector<T> &got, const std::vector<T> &want, double tol = 0) {
  int bad = 0;
  for (size_t i = 0; i < want.size(); ++i)
    if (std::fabs((double)got[i] - (double)want[i]) > tol) ++bad;
  std::printf("  %-10s %s\n", what, bad ? "MISMATCH" : "ok");
  return bad;
}

int main() {
  std::printf("libgsage ABI version %d\n", gsage_version());
  int bad = 0;
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  // ---------------------------------------------------------------- KAT-1: one SAGEConv layer
  const int64_t N = 4, E = 5, Fi = 2, Fo = 3;
  int64_t *ei = to_device<int64_t>({0, 0, 1, 2, 2, /* targets */ 1, 2, 2, 0, 0});
  float *x = to_device<float>({1, 2, 3, 4, 5, 6, 7, 8});
  float *w_l = to_device<float>({1, 0, 0, 1, 1, 1}), *b_l = to_device<float>({1, -1, 0});
  float *w_r = to_device<float>({2, 0, 0, -1, 1, -1});
  int32_t *rowptr = device_alloc<int32_t>(N + 1), *rowptr_t = device_alloc<int32_t>(N + 1);
  int32_t *col = device_alloc<int32_t>(E), *col_t = device_alloc<int32_t>(E);
  float *inv_deg = device_alloc<float>(N);
  const int64_t cap = E / gsage_long_row_threshold() + 1;
  int32_t *lr = device_alloc<int32_t>(cap), *lrt = device_alloc<int32_t>(cap);
  int32_t *n_long = device_alloc<int32_t>(2), *status = device_alloc<int32_t>(1);
  size_t ws_bytes = gsage_csr_workspace_bytes(E, N);
  void *ws = device_alloc<char>(ws_bytes);
  GS_OK(gsage_csr_build(ei, E, N, rowptr, col, rowptr_t, col_t, inv_deg, lr, lrt, cap, n_long, status,
                        ws, ws_bytes, st));
  HIP_OK(hipStreamSynchronize(st));
  std::printf("KAT-1 (4 nodes, duplicate edge, isolated node)\n");
  bad += expect<int32_t>("rowptr", to_host(rowptr, N + 1), {0, 2, 3, 5, 5});
  bad += expect<int32_t>("col", to_host(col, E), {2, 2, 0, 0, 1});
  bad += expect<int32_t>("rowptr_t", to_host(rowptr_t, N + 1), {0, 2, 3, 5, 5});
  bad += expect<int32_t>("col_t", to_host(col_t, E), {1, 2, 2, 0, 0});
  bad += expect<float>("inv_deg", to_host(inv_deg, N), {0.5f, 1.f, 0.5f, 1.f});
  bad += expect<int32_t>("status", to_host(status, 1), {0});

  float *agg = device_alloc<float>(N * Fi), *out = device_alloc<float>(N * Fo);
  GS_OK(gsage_segment_gather(rowptr, col, inv_deg, nullptr, x, nullptr, N, Fi, agg, lr, n_long, st));
  void *wpack = device_alloc<char>(gsage_project_workspace_bytes(Fi, Fo));
  GS_OK(gsage_project_fwd(agg, x, w_l, b_l, w_r, N, Fi, Fo, GSAGE_EPI_NONE, nullptr, 0, 0.f, out, nullptr,
                          wpack, 0, st));
  HIP_OK(hipStreamSynchronize(st));
  bad += expect<float>("agg", to_host(agg, N * Fi), {5, 6, 1, 2, 2, 3, 0, 0});
  bad += expect<float>("out", to_host(out, N * Fo), {8, 3, 10, 8, -3, 2, 13, -4, 4, 15, -9, -1});

  // backward of L = sum(out * G)
  float *G = to_device<float>({1, 0, 2, 0, 1, 0, 1, 1, 1, 2, 0, -1});
  float *gagg = device_alloc<float>(N * Fi), *gx = device_alloc<float>(N * Fi);
  float *gwl = device_alloc<float>(Fo * Fi), *gwr = device_alloc<float>(Fo * Fi), *gb = device_alloc<float>(Fo);
  float *dpre = device_alloc<float>(N * Fo);
  size_t bws_bytes = gsage_project_bwd_workspace_bytes(N, Fi, Fo);
  void *bws = device_alloc<char>(bws_bytes);
  GS_OK(gsage_project_bwd(G, out, nullptr, agg, x, w_l, w_r, inv_deg, N, Fi, Fo, GSAGE_EPI_NONE, 0.f, dpre,
                          gagg, gx, gwl, gb, gwr, bws, bws_bytes, 0, st));
  // grad_agg arrives pre-scaled by inv_deg: gather it over the transposed CSR onto the root path
  GS_OK(gsage_segment_gather(rowptr_t, col_t, nullptr, nullptr, gagg, gx, N, Fi, gx, lrt, n_long + 1, st));
  HIP_OK(hipStreamSynchronize(st));
  bad += expect<float>("grad_b_l", to_host(gb, Fo), {4, 2, 2});
  bad += expect<float>("grad_w_l", to_host(gwl, Fo * Fi), {7, 9, 3, 5, 12, 15});
  bad += expect<float>("grad_w_r", to_host(gwr, Fo * Fi), {20, 24, 8, 10, 0, 2});
  bad += expect<float>("grad_x", to_host(gx, N * Fi), {5, 0, 1, 0, 6, 0, 3, 1});

  // ---------------------------------------------------------------- KAT-5: skip-gram loss
  std::printf("KAT-5 (one pair, one negative, scale 1)\n");
  float *emb = to_device<float>({1, 0, 0, 1});
  int64_t *tg = to_device<int64_t>({0}), *cx = to_device<int64_t>({0}), *ng = to_device<int64_t>({1});
  float *coef = device_alloc<float>(2), *stats = device_alloc<float>(8), *gemb = device_alloc<float>(4);
  float *one = to_device<float>({1.f});
  size_t lws_bytes = gsage_skipgram_workspace_bytes(1);
  void *lws = device_alloc<char>(lws_bytes);
  GS_OK(gsage_skipgram_fwd(emb, 2, 2, tg, cx, ng, 1, 1, 1.0f, 1.0, 1.0, coef, stats, lws, lws_bytes, st));
  GS_OK(gsage_skipgram_bwd(emb, 2, 2, tg, cx, ng, 1, 1, coef, one, gemb, st));
  HIP_OK(hipStreamSynchronize(st));
  const float sig = 1.0f / (1.0f + std::exp(-1.0f));
  const float loss = std::log1p(std::exp(-1.0f)) + std::log(2.0f);
  std::vector<float> s = to_host(stats, 8);
  bad += expect<float>("loss", {s[0]}, {loss}, 1e-6);
  bad += expect<float>("pos_prob", {s[5]}, {sig}, 1e-6);
  bad += expect<float>("neg_prob", {s[6]}, {0.5f}, 1e-6);
  // d/de0 = -(1-sig)*2*e0 (target and context are both row 0) + 0.5*e1 ; d/de1 = 0.5*e0
  bad += expect<float>("grad_emb", to_host(gemb, 4), {-2.f * (1.f - sig), 0.5f, 0.5f, 0.f}, 1e-6);

  std::printf(bad ? "FAILED (%d mismatches)\n" : "all known answers reproduced through the C ABI\n", bad);
  return bad ? 1 : 0;
}
