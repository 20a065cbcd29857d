// Probe: v_mfma_f64_16x16x4_f64 operand/result lane maps and issue rate on gfx950, plus f64 VALU FMA and
// sincos throughput.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void layout_kernel(const double* A, const double* B, double* D) {
  // A is 16x4 row-major (i,k), B is 4x16 row-major (k,j), D 16x16 row-major
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters, double x, unsigned long long* clk = nullptr) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = x + threadIdx.x * 1e-3, b = x - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (clk && threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - t0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters, double x) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = x + i + threadIdx.x * 1e-6;
  double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fma(a[i], m, c);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) sincos_kernel(double* out, int iters, double x) {
  double ph = x + threadIdx.x * 0.37 + blockIdx.x * 0.011, s = 0;
  for (int it = 0; it < iters; ++it) {
    double sn, cs;
    sincos(ph, &sn, &cs);
    s += sn * 0.5 + cs;
    ph += 1.2345;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) sincospi_kernel(double* out, int iters, double x) {
  double ph = x + threadIdx.x * 0.37 + blockIdx.x * 0.011, s = 0;
  for (int it = 0; it < iters; ++it) {
    double sn, cs;
    sincospi(ph, &sn, &cs);
    s += sn * 0.5 + cs;
    ph += 0.12345;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz lds/block %zu regs/block %d\n", p.gcnArchName, p.multiProcessorCount, p.clockRate,
         p.sharedMemPerBlock, p.regsPerBlock);
  // ---- layout
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 7;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
  printf("LAYOUT f64 16x16x4: A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15 : %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  // ---- rates
  int ncu = p.multiProcessorCount;
  double* out; CK(hipMalloc(&out, (size_t)ncu * 8 * 1024 * 8));
  int iters = 2000;
  for (int wpb = 4; wpb <= 8; wpb += 4) {
    for (int bpc = 1; bpc <= 2; ++bpc) {
      int blocks = ncu * bpc, threads = 64 * wpb;
      float ms = time_ms([&] { rate_kernel<8><<<blocks, threads>>>(out, iters, 1.0); }, 5);
      double flops = (double)blocks * wpb * iters * 8 * 2048.0;
      printf("MFMA f64 16x16x4: %d blocks x %d waves, 8 acc: %.3f ms  %.1f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", blocks, wpb, ms,
             flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)wpb * bpc / 4 * iters * 8));
    }
  }
  {
    float ms = time_ms([&] { rate_kernel<2><<<ncu, 256>>>(out, iters, 1.0); }, 5);
    printf("MFMA f64 2 acc (dependent chains), 1 wave/SIMD: %.1f cyc/mfma\n", ms * 1e-3 * 2.4e9 / (iters * 2.0));
    ms = time_ms([&] { rate_kernel<1><<<ncu, 256>>>(out, iters, 1.0); }, 5);
    printf("MFMA f64 1 acc (dependent chain), 1 wave/SIMD: %.1f cyc/mfma\n", ms * 1e-3 * 2.4e9 / (iters * 1.0));
  }
  {
    unsigned long long* clk; CK(hipMalloc(&clk, 16));
    int longit = 200000;
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
      int blocks = ncu * bpc;
      float ms;
      if (bpc <= 2) ms = time_ms([&] { rate_kernel<8><<<blocks, 256>>>(out, longit, 1.0, clk); }, 2);
      else if (bpc == 4) ms = time_ms([&] { rate_kernel<4><<<blocks, 256>>>(out, longit, 1.0, clk); }, 2);
      else ms = time_ms([&] { rate_kernel<2><<<blocks, 256>>>(out, longit, 1.0, clk); }, 2);
      int nacc = bpc <= 2 ? 8 : (bpc == 4 ? 4 : 2);
      unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
      double ghz = (double)h[0] / (double)h[1] * 0.1;
      double flops = (double)blocks * 4 * longit * nacc * 2048.0;
      printf("LONG MFMA f64: %d waves/SIMD, %d acc: %.1f ms %.1f TFLOP/s, in-kernel clock %.2f GHz, %.1f shader-cyc per mfma per SIMD\n", bpc, nacc, ms,
             flops / ms / 1e9, ghz, ms * 1e-3 * ghz * 1e9 / ((double)bpc * longit * nacc));
    }
  }
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    int blocks = ncu * bpc;
    float ms = time_ms([&] { fma_kernel<<<blocks, 256>>>(out, iters, 1.0); }, 5);
    printf("VALU fma f64: %d blocks x 4 waves: %.3f ms %.1f TFLOP/s\n", blocks, ms, (double)blocks * 256 * iters * 16 * 2 / ms / 1e9);
  }
  for (int bpc = 2; bpc <= 8; bpc *= 2) {
    int blocks = ncu * bpc;
    float ms = time_ms([&] { sincos_kernel<<<blocks, 256>>>(out, 500, 1.0); }, 5);
    printf("sincos f64: %d blocks: %.3f ms %.1f G sincos/s\n", blocks, ms, (double)blocks * 256 * 500 / ms / 1e6);
    ms = time_ms([&] { sincospi_kernel<<<blocks, 256>>>(out, 500, 1.0); }, 5);
    printf("sincospi f64: %d blocks: %.3f ms %.1f G sincos/s\n", blocks, ms, (double)blocks * 256 * 500 / ms / 1e6);
  }
  return 0;
}
