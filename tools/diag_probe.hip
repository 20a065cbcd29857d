// Diagnostic (GPU box): cycles and correctness of diag_wave (csrc/eaqhm_ls_chol.h), the one-wave MFMA factorisation +
// inversion of a 16x16 Hermitian positive definite tile.    hipcc --offload-arch=gfx950 -O3 -o tools/diag_probe tools/diag_probe.hip
#include "../eaqhm-analysis-and-synthesis-in-python_amd/csrc/eaqhm_ls_chol.h"
#include <complex>
#include <vector>
#include <random>
using namespace eaqhm;
typedef std::complex<double> cd;

extern "C" __global__ void __launch_bounds__(512) probe(const double* tile, double* out, unsigned long long* cyc, int* fault, int reps) {
  __shared__ double cs[256];
  __shared__ double Wt[2 * TL_TILE], Ld[2 * TL_TILE], dref[16];
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x, lq = lane >> 4, lcol = lane & 15;
  d4 R, I;
  for (int r = 0; r < 4; ++r) { R[r] = tile[2 * ((lq + 4 * r) * 16 + lcol)]; I[r] = tile[2 * ((lq + 4 * r) * 16 + lcol) + 1]; }
  if (lane < 16) dref[lane] = tile[2 * (lane * 16 + lane)];
  __builtin_amdgcn_wave_barrier();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < reps; ++k) diag_wave(R, I, cs, Wt, Wt + TL_TILE, Ld, Ld + TL_TILE, true, dref, 16, fault);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  for (int q = lane; q < 2 * TL_TILE; q += 64) { out[q] = Wt[q]; out[2 * TL_TILE + q] = Ld[q]; }
}

int main() {
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd;
  std::vector<cd> A(40 * 16), T(256);
  for (auto& v : A) v = cd(nd(g), nd(g));
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { cd s = 0; for (int t = 0; t < 40; ++t) s += std::conj(A[t * 16 + i]) * A[t * 16 + j]; T[i * 16 + j] = s; }
  double *dT, *dO; unsigned long long* dC; int* dF;
  hipMalloc(&dT, 512 * 8); hipMalloc(&dO, 4 * TL_TILE * 8); hipMalloc(&dC, 8); hipMalloc(&dF, 4);
  hipMemcpy(dT, T.data(), 512 * 8, hipMemcpyHostToDevice); hipMemset(dF, 0, 4);
  std::vector<double> O(4 * TL_TILE);
  for (int reps : {1, 1, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, dT, dO, dC, dF, reps);
    hipDeviceSynchronize();
    unsigned long long c; int f;
    hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, dF, 4, hipMemcpyDeviceToHost);
    hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost);
    // W^H at [k*TL_LD + j] = conj(W[j][k]);  L[i][j] at [i*TL_LD + j]
    double eL = 0, eW = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      cd s = 0, w = 0;
      for (int k = 0; k < 16; ++k) {
        cd Lik(O[2 * TL_TILE + i * TL_LD + k], O[3 * TL_TILE + i * TL_LD + k]), Ljk(O[2 * TL_TILE + j * TL_LD + k], O[3 * TL_TILE + j * TL_LD + k]);
        s += Lik * std::conj(Ljk);
        cd Wik(O[k * TL_LD + i], -O[TL_TILE + k * TL_LD + i]);   // W[i][k]
        w += Wik * cd(O[2 * TL_TILE + k * TL_LD + j], O[3 * TL_TILE + k * TL_LD + j]);   // (W L)[i][j]
      }
      eL = fmax(eL, std::abs(s - T[i * 16 + j]));
      eW = fmax(eW, std::abs(w - cd(i == j ? 1.0 : 0.0, 0.0)));
    }
    printf("reps %d: %.0f cycles per call, |L L^H - T| = %.2e, |W L - I| = %.2e, faults %d\n", reps, (double)c / reps, eL, eW, f);
  }
  return 0;
}
