// Probe (diagnostics, not product code) for v_mfma_f32_4x4x1_16b_f32 on gfx950:
//   1. operand layout and the CBSZ/ABID A-broadcast, checked against a host model;
//   2. issue rate: one dependent accumulator chain vs 2 / 4 independent chains, against 16x16x4;
//   3. does it co-issue with VALU FMAs of the partner waves on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma4x4_probe.hip -o /tmp/p4 && /tmp/p4
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CBSZ, int ABID>
__global__ void sem(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = acc[i];
}

static int check(const char* name, int cbsz, int abid, const std::vector<float>& a, const std::vector<float>& b,
                 const std::vector<float>& d) {
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int blk = l >> 2, j = l & 3;
    const int grp = 1 << cbsz;
    const int src_blk = cbsz ? (blk / grp) * grp + abid : blk;
    for (int i = 0; i < 4; ++i) {
      const float want = a[4 * src_blk + i] * b[4 * blk + j];
      if (want != d[l * 4 + i]) ++bad;
    }
  }
  printf("semantics %-16s cbsz=%d abid=%d : %s (%d mismatches)\n", name, cbsz, abid, bad ? "MISMATCH" : "ok", bad);
  return bad;
}

// MODE 0: one chain of 4x4x1; 1: two chains; 2: four chains; 3: 16x16x4 (4 chains); +8: VALU partner waves active
template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, int iters, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  const float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if ((MODE & 7) == 0) {
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 3, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a0, 3, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 3, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a0, 3, 0, 0);
        } else if ((MODE & 7) == 1) {
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 3, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 3, 0, 0);
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 3, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 3, 0, 0);
        } else if ((MODE & 7) == 2) {
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 3, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a1, 3, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 3, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, a3, 3, 0, 0);
        } else {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a3, 0, 0, 0);
        }
      }
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (MODE & 8) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], y, x);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) r += v[j];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
static void run_rate(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  rate<MODE><<<blocks, 512>>>(out, 100, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<MODE><<<blocks, 512>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double mfma = (double)iters * 32;
  // s_memtime counts at 100 MHz on gfx9: report wall time per MFMA instead and the implied flop rate
  const double flop = ((MODE & 7) == 3 ? 2048.0 : 512.0) * mfma * 4 * blocks;
  printf("%-34s %8.3f ms  %7.2f ns/MFMA/wave  %7.1f TFLOP/s (matrix waves only)  memtime wave0 %llu wave4 %llu\n", name,
         ms, ms * 1e6 / mfma, flop / (ms * 1e-3) / 1e12, h[0], h[4]);
}

int main() {
  std::vector<float> a(64), b(64), d(256);
  for (int i = 0; i < 64; ++i) {
    a[i] = 1.0f + i;
    b[i] = 100.0f + 3 * i;
  }
  float *da, *db, *dd;
  hipMalloc(&da, 256);
  hipMalloc(&db, 256);
  hipMalloc(&dd, 1024);
  hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
  int bad = 0;
#define SEM(C, A)                                             \
  sem<C, A><<<1, 64>>>(da, db, dd);                           \
  hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);       \
  bad += check("4x4x1_16b", C, A, a, b, d);
  SEM(0, 0) SEM(3, 0) SEM(3, 1) SEM(3, 5) SEM(2, 1) SEM(1, 1) SEM(4, 9)
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 64);
  run_rate<0>("4x4x1 one chain", out, cyc);
  run_rate<1>("4x4x1 two chains", out, cyc);
  run_rate<2>("4x4x1 four chains", out, cyc);
  run_rate<3>("16x16x4 four chains", out, cyc);
  run_rate<8>("4x4x1 one chain + VALU waves", out, cyc);
  run_rate<10>("4x4x1 four chains + VALU waves", out, cyc);
  run_rate<11>("16x16x4 four chains + VALU waves", out, cyc);
  printf(bad ? "PROBE: semantic mismatches\n" : "PROBE: semantics as modelled\n");
  return 0;
}
