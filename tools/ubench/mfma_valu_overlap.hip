// Microbenchmark (diagnostics, not product code): do f32-input MFMA and f32 VALU FMA overlap on one
// SIMD?  Waves 0-3 of a workgroup (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD partners)
// run a VALU FMA loop.  Compare: MFMA alone, VALU alone, both together; f32 MFMA vs bf16 MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>  // bit0: MFMA waves active, bit1: VALU waves active, bit2: bf16 MFMA instead of f32
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (MODE & 1) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
      bf16x8 bx = {1, 2, 3, 4, 5, 6, 7, 8}, by = {8, 7, 6, 5, 4, 3, 2, 1};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (MODE & 4) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a3, 0, 0, 0);
          } else {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
          }
        }
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else {
    if (MODE & 2) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
      const float m = 0.999f, c = 1e-3f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], m, c);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) r += v[j];
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
float run(float* d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512>>>(d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, 512>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int it = 20000;
  // per iteration: MFMA waves issue 16 MFMA (f32: 16*32 = 512 cycles; bf16 16x16x32: 16*16 = 256),
  // VALU waves issue 32 v_fma_f32 (if 4 cycles each alone: 128 cycles)
  printf("f32 MFMA only     : %.3f ms\n", run<1>(d, it));
  printf("VALU only         : %.3f ms\n", run<2>(d, it));
  printf("f32 MFMA + VALU   : %.3f ms\n", run<3>(d, it));
  printf("bf16 MFMA only    : %.3f ms\n", run<5>(d, it));
  printf("bf16 MFMA + VALU  : %.3f ms\n", run<7>(d, it));
  printf("VALU x4 iters only: %.3f ms\n", run<2>(d, it * 4));
  return 0;
}
