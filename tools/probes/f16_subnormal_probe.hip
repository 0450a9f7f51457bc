// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs, and does the f32->f16 conversion
// produce them?  Build: hipcc --offload-arch=gfx950 -O3 f16_subnormal_probe.hip -o f16_subnormal_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float *out, unsigned *bits) {
  f16x8 a, b;
  const float tiny = 9.5367431640625e-07f;   // 2^-20: subnormal in fp16 (bits 0x0010)
  _Float16 h = (_Float16)(tiny * (threadIdx.x >= 0 ? 1.f : 2.f));
  for (int i = 0; i < 8; ++i) { a[i] = h; b[i] = (_Float16)1024.f; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = c[0];
    union { _Float16 f; unsigned short u; } v; v.f = h;
    bits[0] = v.u;
    // residual split of a value whose low part is subnormal
    const float x = 0.01f;
    _Float16 x1 = (_Float16)x;
    const float r = x - (float)x1;
    _Float16 x2 = (_Float16)r;
    union { _Float16 f; unsigned short u; } w; w.f = x2;
    bits[1] = w.u;
    out[1] = r; out[2] = (float)x2;
  }
}
int main() {
  float *o; unsigned *b;
  hipMalloc(&o, 64); hipMalloc(&b, 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, o, b);
  float ho[4]; unsigned hb[4];
  hipMemcpy(ho, o, 16, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 16, hipMemcpyDeviceToHost);
  printf("mfma(sub 2^-20 x 1024, 16 k) = %g (expect 0.015625 if honoured, 0 if flushed)\n", ho[0]);
  printf("cvt(2^-20) bits = 0x%04x (expect 0x0010)\n", hb[0]);
  printf("residual of 0.01: r=%g x2=%g bits=0x%04x\n", ho[1], ho[2], hb[1]);
  return 0;
}
