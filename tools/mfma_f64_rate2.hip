// Probe 2: what sets the issue rate of v_mfma_f64_16x16x4_f64 on gfx950 — operand reuse, accumulator order, VALU between.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_rate2 tools/mfma_f64_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

// MODE 0: 8 acc, same a, b         MODE 1: 8 acc, distinct a[i], b[i]     MODE 2: 2 x 4 outer product (a[m], b[c])
// MODE 3: each accumulator twice in a row (acc0, acc0, acc1, acc1, ...), distinct operands
// MODE 4: MODE 1 + one independent v_fma_f64 after every MFMA     MODE 5: MODE 2 with 2 x 3 x 3 (18 acc, three-product pattern)
// MODE 6: 8 acc, same a, distinct b   MODE 7: 8 acc, distinct a, same b
template <int MODE>
__global__ void __launch_bounds__(512) rate(double* out, int iters, double x, unsigned long long* clk) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  d4 acc[18];
  for (int i = 0; i < 18; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[8], b[8], f = x;
  for (int i = 0; i < 8; ++i) { a[i] = x + threadIdx.x * 1e-3 + i; b[i] = x - threadIdx.x * 1e-3 - i; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MF(a[0], b[0], acc[i]);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MF(a[i], b[i], acc[i]);
    } else if (MODE == 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) MF(a[m], b[c], acc[m * 4 + c]);
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { MF(a[i], b[i], acc[i]); MF(a[i + 4], b[i + 4], acc[i]); }
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { MF(a[i], b[i], acc[i]); f = fma(f, 1.0000001, 1e-9); }
    } else if (MODE == 5) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          MF(a[m], b[c], acc[(m * 3 + c) * 3]);
          MF(a[m + 2], b[c + 3], acc[(m * 3 + c) * 3 + 1]);
          MF(a[m + 4], b[c + 5], acc[(m * 3 + c) * 3 + 2]);
        }
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MF(a[0], b[i], acc[i]);
    } else if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 8; ++i) MF(a[i], b[0], acc[i]);
    }
    // keep the operands "changing" so that nothing is hoisted or merged (cheap scalar-ish update once per trip)
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  double s = f;
  for (int i = 0; i < 18; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (clk && threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int MODE>
static int run(const char* name, int per_trip, double* out, unsigned long long* clk, int ncu) {
  const int iters = 40000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    int threads = (wps == 4) ? 512 : 256 * wps, blocks = (wps == 4) ? 2 * ncu : ncu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<MODE><<<blocks, threads>>>(out, 100, 1.0, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<MODE><<<blocks, threads>>>(out, iters, 1.0, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles at the nominal 2.4 GHz from the wall time of the launch (every CU busy, 4 SIMDs x wps waves each)
    printf("%-46s %d waves/SIMD: %6.1f cyc per MFMA per SIMD at 2.4 GHz, %.1f TFLOP/s\n", name, wps,
           ms * 1e-3 * 2.4e9 / ((double)wps * iters * per_trip), (double)ncu * 4 * wps * iters * per_trip * 2048.0 / ms / 1e9);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, (size_t)ncu * 2 * 1024 * 8));
  unsigned long long* clk; CK(hipMalloc(&clk, 16));
  run<0>("0: 8 acc, same a, same b", 8, out, clk, ncu);
  run<1>("1: 8 acc, distinct a[i], b[i]", 8, out, clk, ncu);
  run<2>("2: 2 x 4 outer product", 8, out, clk, ncu);
  run<3>("3: every accumulator twice in a row", 8, out, clk, ncu);
  run<4>("4: as 1 + a v_fma_f64 after every MFMA", 8, out, clk, ncu);
  run<5>("5: 2 x 3 tiles x 3 products (18 acc)", 18, out, clk, ncu);
  run<6>("6: 8 acc, same a, distinct b", 8, out, clk, ncu);
  run<7>("7: 8 acc, distinct a, same b", 8, out, clk, ncu);
  return 0;
}
