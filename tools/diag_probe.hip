// Diagnostic (GPU box): cycles and correctness of diag_wave (csrc/eaqhm_ls_chol.h), the one-wave MFMA factorisation +
// inversion of a 16x16 Hermitian positive definite tile.    hipcc --offload-arch=gfx950 -O3 -o tools/diag_probe tools/diag_probe.hip
#include "../eaqhm-analysis-and-synthesis-in-python_amd/csrc/eaqhm_ls_chol.h"
#include <complex>
#include <vector>
#include <random>
using namespace eaqhm;
typedef std::complex<double> cd;

extern "C" __global__ void __launch_bounds__(512) probe(const double* tile, double* out, unsigned long long* cyc, int* fault, int reps) {
  __shared__ double post[DGP_DOUBLES], dumpD[128], zs[256];
  __shared__ double Wt[2 * TL_TILE], Ld[2 * TL_TILE], dref[16];
  __shared__ int flag;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63, lq = lane >> 4, lcol = lane & 15;
  d4 R, I;
  for (int r = 0; r < 4; ++r) { R[r] = tile[2 * ((lq + 4 * r) * 16 + lcol)]; I[r] = tile[2 * ((lq + 4 * r) * 16 + lcol) + 1]; }
  if (threadIdx.x < 16) dref[threadIdx.x] = tile[2 * (threadIdx.x * 16 + threadIdx.x)];
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < reps; ++k) {
    if (wave == 0) diag_D(R, I, post, &flag, 16 * k, dumpD, Ld, Ld + TL_TILE, k == 0);
    else if (wave == 1) diag_Z(post, &flag, 16 * k, zs, Wt, Wt + TL_TILE, dref, 16, fault);
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  for (int q = threadIdx.x; q < 2 * TL_TILE; q += 512) { out[q] = Wt[q]; out[2 * TL_TILE + q] = Ld[q]; }
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
      if (false) eL = fmax(eL, std::abs(s - T[i * 16 + j]));
      eW = fmax(eW, std::abs(w - cd(i == j ? 1.0 : 0.0, 0.0)));
    }
    // W W^H must be T^-1:  check  W^H W T = I  (W = L^-1  =>  T^-1 = W^H W)
    double eT = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      cd acc = 0;
      for (int k = 0; k < 16; ++k) for (int m = 0; m < 16; ++m) {
        cd Wki(O[i * TL_LD + k], -O[TL_TILE + i * TL_LD + k]);     // W[k][i]
        cd Wkm(O[m * TL_LD + k], -O[TL_TILE + m * TL_LD + k]);     // W[k][m]
        acc += std::conj(Wki) * Wkm * T[m * 16 + j];
      }
      eT = fmax(eT, std::abs(acc - cd(i == j ? 1.0 : 0.0, 0.0)));
    }
    // row 15 of L (what the RHS extraction reads) against the direct formula L[15][j] = conj((W^-1)...): use L = T W^H
    double eR = 0;
    for (int j = 0; j < 14; ++j) {
      cd acc = 0;
      for (int k = 0; k < 16; ++k) acc += T[15 * 16 + k] * cd(O[k * TL_LD + j], O[TL_TILE + k * TL_LD + j]);   // (T W^H)[15][j], W^H[k][j]
      eR = fmax(eR, std::abs(acc - cd(O[2 * TL_TILE + 15 * TL_LD + j], O[3 * TL_TILE + 15 * TL_LD + j])));
    }
    printf("reps %d: %.0f cycles per call (two-wave pipeline), |W^H W T - I| = %.2e, |L row 15 - (T W^H) row 15| = %.2e, faults %d\n", reps, (double)c / reps, eT, eR, f);
  }
  return 0;
}
