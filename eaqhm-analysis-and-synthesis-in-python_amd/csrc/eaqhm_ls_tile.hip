// eaqhm_ls_tile.hip — the per-frame LS entirely on chip: Gramian AND factorisation on the FP64 matrix cores,
// the system matrix never leaves the register file (gfx950).
//
// A workgroup of 512 threads (8 waves, 2 per SIMD) owns one frame at a time (atomic frame queue).
//
// Unknown ordering.  The basis columns are cut into nb blocks of 16 ([negative | DC | positive | signal | 0-pad]).
// Unknowns are ordered block by block, amplitudes (alpha=0) then slopes (alpha=1) of each block, except that
// the amplitude part of the LAST block comes last.  The signal window is the last real column of that block,
// so its row of the Hermitian system is the last real row: factorising the matrix WITH that row/column leaves
// conj(L^-1 rhs) in it — the forward substitution costs nothing.  System tile (P,Q) = G_{alpha_P+alpha_Q}[I_P][I_Q]
// with G_p = X^H diag(w^2 n^p) X: every 16x16 complex system tile is one MFMA accumulation over time, owned by
// one wave (tile x = P(P+1)/2+Q -> wave x%8, slot x/8) from the first sample to the last back-substitution step.
//
//   A1     slot windows, gap fill, running sums (eaqhm_ls_common.h)                      -> global scratch
//   A3+B   chunks of 16 sample pairs built in LDS by all threads, contracted by MFMA (see eaqhm_ls_mfma.hip)
//   C      right-looking tile Cholesky: the diagonal tile is factorised AND inverted in registers by its owner
//          (wave shuffles), panel tiles are multiplied by the inverse (MFMA) and published in LDS, trailing
//          tiles are updated from LDS operands (MFMA).  Two barriers per panel; no global memory traffic.
//   C'     back substitution from the L tiles still sitting in the owners' registers; z, x vectors in LDS
//   D      frequency mismatch, acceptance, record row (eaqhm_ls_common.h)
//
// Frames with more than 6 column blocks (Kc > 95) do not fit the register budget and are left to
// eaqhm_ls_mfma_kernel (same Gramian, factorisation through scratch memory).
#include "eaqhm_ls_common.h"

namespace eaqhm {

typedef double d4 __attribute__((ext_vector_type(4)));

#define TL_THREADS 512
#define TL_WAVES 8
#define TL_NS 10        // system tiles per wave: 80 >= 78 = nt(nt+1)/2 for nt = 12 (nb = 6)
#define TL_NBMAX 6
#define TL_NTMAX 12
#define TL_LD 17        // tile row stride in LDS (doubles): conflict-free transposing stores
#define TL_TILE (16 * TL_LD)

__device__ inline void sys_tile_of(int x, int& P, int& Q) {
  P = (int)((sqrtf(8.0f * (float)x + 1.0f) - 1.0f) * 0.5f);
  while ((P + 1) * (P + 2) / 2 <= x) ++P;
  while (P * (P + 1) / 2 > x) --P;
  Q = x - P * (P + 1) / 2;
}
// position in the unknown ordering -> (column block, alpha)
__device__ inline void block_of(int P, int nt, int& I, int& alpha) {
  if (P < nt - 2) { I = P >> 1; alpha = P & 1; }
  else { I = (nt >> 1) - 1; alpha = (P == nt - 2) ? 1 : 0; }
}

__device__ inline double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// Cholesky factor L and its inverse W of a 16x16 Hermitian positive definite tile held in MFMA accumulator
// layout (lane: col = l&15, rows (l>>4)+4r).  By symmetry the same registers read as
// D[i][k] = conj(acc[r]) with i = l&15, k = (l>>4)+4r.  Outputs in that (row i, column k) layout.
__device__ inline void factor_diag(const d4& aR, const d4& aI, double (&Lr)[4], double (&Li)[4], double (&Wr)[4],
                                   double (&Wi)[4], int lane) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    Lr[m] = aR[m]; Li[m] = -aI[m];
    Wr[m] = (i == kq + 4 * m) ? 1.0 : 0.0; Wi[m] = 0.0;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int mj = j >> 2, qj = j & 3;
    double piv = shfl_d(Lr[mj], j + 16 * qj);
    piv = (piv > 0.0) ? piv : 1.0;  // only the RHS position of the last tile can get here (residual energy ~ 0)
    const double dinv = 1.0 / sqrt(piv);
    if (kq == qj) {  // column j: scale below the diagonal, clean above
      if (i > j) { Lr[mj] *= dinv; Li[mj] *= dinv; }
      else if (i == j) { Lr[mj] = piv * dinv; Li[mj] = 0.0; }
      else { Lr[mj] = 0.0; Li[mj] = 0.0; }
    }
    const double lijr = shfl_d(Lr[mj], i + 16 * qj), liji = shfl_d(Li[mj], i + 16 * qj);  // L[i][j]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int k = kq + 4 * m;
      const double lkr = shfl_d(Lr[mj], k + 16 * qj), lki = shfl_d(Li[mj], k + 16 * qj);  // L[k][j]
      if (k > j && i >= k) {  // D[i][k] -= L[i][j] conj(L[k][j])
        Lr[m] -= lijr * lkr + liji * lki;
        Li[m] -= liji * lkr - lijr * lki;
      }
      // inverse by forward elimination on [L | I]: row j /= L[j][j]; rows i > j -= L[i][j] * row j
      const double zr = shfl_d(Wr[m], j + 16 * kq) * dinv, zi = shfl_d(Wi[m], j + 16 * kq) * dinv;
      if (i == j) { Wr[m] = zr; Wi[m] = zi; }
      else if (i > j) {
        Wr[m] -= lijr * zr - liji * zi;
        Wi[m] -= lijr * zi + liji * zr;
      }
    }
  }
}

extern "C" __global__ void __launch_bounds__(TL_THREADS) eaqhm_ls_tile_kernel(LsArgs A, int TS, int ldx_max) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, nt_thr = TL_THREADS;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- LDS carve-up: region U is used as the basis chunk in the Gramian phase and as tile storage afterwards
  double* U = lds;
  double* Xre = U;
  double* Xim = Xre + (size_t)TS * ldx_max;
  double* Wp = Xim + (size_t)TS * ldx_max;  // 3*TS
  double* PanR = U;                                  // [NT][TILE]  published panel tiles, [k][row]
  double* PanI = PanR + TL_NTMAX * TL_TILE;
  double* WtR = PanI + TL_NTMAX * TL_TILE;           // [NT][TILE]  (W^H)[k][j] of every diagonal tile
  double* WtI = WtR + TL_NTMAX * TL_TILE;
  double* TmpR = WtI + TL_NTMAX * TL_TILE;           // [8][TILE]   per-wave transposition buffer
  double* TmpI = TmpR + TL_WAVES * TL_TILE;
  double* LdR = TmpI + TL_WAVES * TL_TILE;           // [TILE]      last diagonal factor, [row][col]
  double* LdI = LdR + TL_TILE;
  double* zv = LdI + TL_TILE;                        // 2*16*NT
  double* xv = zv + 2 * 16 * TL_NTMAX;               // 2*16
  const size_t usize_c = (size_t)(xv + 32 - U);
  const size_t usize_g = (size_t)2 * TS * ldx_max + 3 * TS;
  double* rho = U + (usize_c > usize_g ? usize_c : usize_g);  // 2*nmax
  double* xs = rho + 2 * A.nmax;                     // 2*Mmax
  double* sh = xs + 4 * A.Kcmax;                     // 16
  int* shi = (int*)(sh + 12);

  double* Qs = A.scratch + (size_t)blockIdx.x * A.scratch_stride;
  double* Rs = Qs + (size_t)(A.Nmax + 1) * A.nmax;
  const bool seeds = (A.mode == 1) && A.any_seed && (*A.any_seed != 0);
  const int PE = TS / 2;
  const int lcol = lane & 15, lq = lane >> 4;

  for (;;) {
    if (tid == 0) shi[0] = atomicAdd(A.work_counter, 1);
    __syncthreads();
    const int f = shi[0];
    __syncthreads();
    if (f >= A.n_frames) break;
    const int n = (A.mode == 0) ? A.frame_K[f] : A.ncol[f];
    const int Kc = 2 * n + 1, C1 = Kc + 1;
    const int nb = (C1 + 15) >> 4;
    if (nb > TL_NBMAX) continue;  // left to eaqhm_ls_mfma_kernel
    const int c = A.frame_c[f], wl = A.frame_wl[f], inst = A.frame_inst[f];
    const int N = 2 * wl + 1, mid = wl;
    const int C1p = nb << 4;
    const int ldx = C1p + ((nb & 1) ? 0 : 16);
    const int nt = 2 * nb, ntiles = nt * (nt + 1) / 2;
    const int is = Kc - 16 * (nb - 1);  // position of the signal column inside the last block (1..15)
    const double f0 = (A.mode == 0) ? A.frame_f0[f] : A.f0_stale;
    const int* mycols = (A.mode == 1) ? (A.cols + (size_t)f * A.Kmax) : nullptr;
    const int npairs = mid + 1;

    // region U may hold tiles of the previous frame: make the basis chunk finite and its padding zero
    for (int q = tid; q < 2 * TS * ldx_max; q += nt_thr) Xre[q] = 0.0;
    if (A.mode == 1) fill_columns(A, Qs, Rs, rho, mycols, n, N, mid, c, wl, seeds);
    __syncthreads();

    d4 accR[TL_NS], accI[TL_NS];
    int tP[TL_NS], tQ[TL_NS];
    bool live[TL_NS];
#pragma unroll
    for (int sl = 0; sl < TL_NS; ++sl) {
      accR[sl] = (d4){0, 0, 0, 0};
      accI[sl] = (d4){0, 0, 0, 0};
      const int x = sl * TL_WAVES + wave;
      live[sl] = x < ntiles;
      int P = 0, Q = 0;
      sys_tile_of(live[sl] ? x : 0, P, Q);
      tP[sl] = P; tQ[sl] = Q;
    }

    // ================= Gramian =================
    for (int e0 = 0; e0 < npairs; e0 += PE) {
#pragma clang loop unroll(disable)
      for (int idx = tid; idx < PE * n; idx += nt_thr) {
        const int el = idx / n, j = idx - el * n, e = e0 + el;
        if (e >= npairs) continue;
        const int u = e - 1, v = N - 1 - e;
        double su = 0, cu = 1, sv, cv;
        double* xr = Xre + (2 * el) * ldx;
        double* xi = Xim + (2 * el) * ldx;
        if (A.mode == 1) {
          sincos_cw((2.0 * M_PI * Qs[(size_t)(u + 1) * n + j]) / A.fs, &su, &cu);
          sincos_cw((2.0 * M_PI * Qs[(size_t)(v + 1) * n + j]) / A.fs, &sv, &cv);
          const double pr = rho[2 * j], pi = rho[2 * j + 1];
          if (u >= 0) {
            const double ru = Rs[(size_t)u * n + j], rv1 = Rs[(size_t)(v + 1) * n + j];
            xr[n + 1 + j] = ru * cu;               xi[n + 1 + j] = ru * su;
            xr[j] = rv1 * (cv * pr - sv * pi);     xi[j] = rv1 * (cv * pi + sv * pr);
          }
          const double rv = Rs[(size_t)v * n + j], ru1 = Rs[(size_t)(u + 1) * n + j];
          xr += ldx; xi += ldx;
          xr[n + 1 + j] = rv * cv;                 xi[n + 1 + j] = rv * sv;
          xr[j] = ru1 * (cu * pr - su * pi);       xi[j] = ru1 * (cu * pi + su * pr);
        } else {
          const double fk = (double)(j + 1) * f0;
          if (u >= 0) {
            sincos_cw(((double)(u - mid) * 2.0 * M_PI * fk) / A.fs, &su, &cu);
            xr[n + 1 + j] = cu; xi[n + 1 + j] = su; xr[j] = cu; xi[j] = -su;
          }
          sincos_cw(((double)(v - mid) * 2.0 * M_PI * fk) / A.fs, &sv, &cv);
          xr += ldx; xi += ldx;
          xr[n + 1 + j] = cv; xi[n + 1 + j] = sv; xr[j] = cv; xi[j] = -sv;
        }
      }
#pragma clang loop unroll(disable)
      for (int row = tid; row < TS; row += nt_thr) {
        const int e = e0 + (row >> 1);
        const int t = (row & 1) ? (N - 1 - e) : (e - 1);
        double w0 = 0.0, sval = 0.0;
        if (e < npairs && t >= 0) {
          double w = window_value(A.mode == 0, t, N);
          w0 = w * w;
          sval = A.s[(size_t)(c - wl) + t];
        }
        const double nn = (double)(t - mid);
        Wp[row] = w0; Wp[TS + row] = w0 * nn; Wp[2 * TS + row] = w0 * nn * nn;
        Xre[row * ldx + n] = 1.0;   Xim[row * ldx + n] = 0.0;
        Xre[row * ldx + Kc] = sval; Xim[row * ldx + Kc] = 0.0;
      }
      __syncthreads();
      const int lbase = lq * ldx + lcol;
#pragma unroll
      for (int sl = 0; sl < TL_NS; ++sl) {
        if (!live[sl]) continue;
        int Ia, aa, Ib, ab;
        block_of(tP[sl], nt, Ia, aa);
        block_of(tQ[sl], nt, Ib, ab);
        const double* wrow = Wp + (aa + ab) * TS + lq;
        const double* pAr = Xre + lbase + 16 * Ia;
        const double* pAi = Xim + lbase + 16 * Ia;
        const double* pBr = Xre + lbase + 16 * Ib;
        const double* pBi = Xim + lbase + 16 * Ib;
#pragma clang loop unroll(disable)
        for (int ks = 0; ks < TS / 4; ++ks) {
          const int ro = 4 * ks * ldx;
          const double aR = pAr[ro], aI = pAi[ro];
          const double w = wrow[4 * ks];
          const double bR = w * pBr[ro], bI = w * pBi[ro];
          accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, accR[sl], 0, 0, 0);
          accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, accR[sl], 0, 0, 0);
          accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bI, accI[sl], 0, 0, 0);
          accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, bR, accI[sl], 0, 0, 0);
        }
      }
      __syncthreads();
    }

    // ---- neutralise dummy unknowns: in the last column block the positions >= is of the slope part (the
    // signal column's slope + padding) and the positions > is of the amplitude part (padding) get an identity
    // row/column; position `is` of the amplitude part is the RHS row/column and stays.
#pragma unroll
    for (int sl = 0; sl < TL_NS; ++sl) {
      if (!live[sl]) continue;
      const int P = tP[sl], Q = tQ[sl];
      if (P < nt - 2) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = lq + 4 * r, col = lcol;
        const bool drow = (P == nt - 2) ? (row >= is) : (row > is);
        const bool dcol = (Q == nt - 2) ? (col >= is) : ((Q == nt - 1) ? (col > is) : false);
        if (drow || dcol) {
          accR[sl][r] = (P == Q && row == col) ? 1.0 : 0.0;
          accI[sl][r] = 0.0;
        }
      }
    }

    // ================= factorisation =================
    for (int jb = 0; jb < nt; ++jb) {
      const int xd = jb * (jb + 1) / 2 + jb;
      if (wave == (xd & 7)) {
        const int sd = xd >> 3;
#pragma unroll
        for (int sl = 0; sl < TL_NS; ++sl) {
          if (sl != sd) continue;
          double Lr[4], Li[4], Wr[4], Wi[4];
          factor_diag(accR[sl], accI[sl], Lr, Li, Wr, Wi, lane);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int k = lq + 4 * m;  // column of L / W, row index is lcol
            WtR[jb * TL_TILE + k * TL_LD + lcol] = Wr[m];
            WtI[jb * TL_TILE + k * TL_LD + lcol] = -Wi[m];
            LdR[lcol * TL_LD + k] = Lr[m];
            LdI[lcol * TL_LD + k] = Li[m];
          }
        }
      }
      __syncthreads();
      // ---- panel tiles (P > jb, Q == jb): X = T W^H, published as Pan[P][k][row]
#pragma unroll
      for (int sl = 0; sl < TL_NS; ++sl) {
        if (!live[sl] || tQ[sl] != jb || tP[sl] == jb) continue;
        double* tr = TmpR + wave * TL_TILE;
        double* ti = TmpI + wave * TL_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // T[row = lq+4r][col = lcol] -> tmp[k = col][i = row]
          tr[lcol * TL_LD + lq + 4 * r] = accR[sl][r];
          ti[lcol * TL_LD + lq + 4 * r] = accI[sl][r];
        }
        __builtin_amdgcn_wave_barrier();
        d4 xr = (d4){0, 0, 0, 0}, xi = (d4){0, 0, 0, 0};
        const double* wr = WtR + jb * TL_TILE;
        const double* wi = WtI + jb * TL_TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int o = (4 * ks + lq) * TL_LD + lcol;
          const double aR = tr[o], aI = ti[o], bR = wr[o], bI = wi[o];
          xr = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, xr, 0, 0, 0);
          xr = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, bI, xr, 0, 0, 0);
          xi = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bI, xi, 0, 0, 0);
          xi = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bR, xi, 0, 0, 0);
        }
        accR[sl] = xr; accI[sl] = xi;  // the finished L tile stays here for the back substitution
        double* pr = PanR + tP[sl] * TL_TILE;
        double* pi = PanI + tP[sl] * TL_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[lcol * TL_LD + lq + 4 * r] = xr[r];
          pi[lcol * TL_LD + lq + 4 * r] = xi[r];
        }
      }
      __syncthreads();
      // ---- trailing tiles (P >= Q > jb): T -= L[P][jb] L[Q][jb]^H
#pragma unroll
      for (int sl = 0; sl < TL_NS; ++sl) {
        if (!live[sl] || tQ[sl] <= jb) continue;
        const double* ar = PanR + tP[sl] * TL_TILE;
        const double* ai = PanI + tP[sl] * TL_TILE;
        const double* br = PanR + tQ[sl] * TL_TILE;
        const double* bi = PanI + tQ[sl] * TL_TILE;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int o = (4 * ks + lq) * TL_LD + lcol;
          const double aR = ar[o], aI = ai[o], lR = br[o], lI = bi[o];
          accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aR, lR, accR[sl], 0, 0, 0);
          accR[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, lI, accR[sl], 0, 0, 0);
          accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, lI, accI[sl], 0, 0, 0);
          accI[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aI, lR, accI[sl], 0, 0, 0);
        }
      }
      // (the barrier after the next diagonal step orders these LDS reads before the next panel's writes)
    }
    __syncthreads();

    // ================= back substitution  L^H x = y,  y = conj(row `is` of the last tile row) =================
#pragma unroll
    for (int sl = 0; sl < TL_NS; ++sl) {
      if (!live[sl] || tP[sl] != nt - 1 || tQ[sl] == nt - 1) continue;
      if (lq == (is & 3)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r == (is >> 2)) {
            zv[2 * (16 * tQ[sl] + lcol)] = accR[sl][r];
            zv[2 * (16 * tQ[sl] + lcol) + 1] = -accI[sl][r];
          }
      }
    }
    if (tid < 16) {
      const bool ok = tid < is;
      zv[2 * (16 * (nt - 1) + tid)] = ok ? LdR[is * TL_LD + tid] : 0.0;
      zv[2 * (16 * (nt - 1) + tid) + 1] = ok ? -LdI[is * TL_LD + tid] : 0.0;
    }
    for (int q = tid; q < 4 * Kc; q += nt_thr) xs[q] = 0.0;
    __syncthreads();
    for (int P = nt - 1; P >= 0; --P) {
      if (tid < 16) {  // x_P = W_PP^H z_P : row tid of (W^H)
        const double* wr = WtR + P * TL_TILE + tid * TL_LD;
        const double* wi = WtI + P * TL_TILE + tid * TL_LD;
        double xr = 0, xi = 0;
        for (int k = tid; k < 16; ++k) {
          const double zr = zv[2 * (16 * P + k)], zi = zv[2 * (16 * P + k) + 1];
          xr += wr[k] * zr - wi[k] * zi;
          xi += wr[k] * zi + wi[k] * zr;
        }
        xv[2 * tid] = xr; xv[2 * tid + 1] = xi;
        int I, alpha;
        block_of(P, nt, I, alpha);
        const int col = 16 * I + tid;
        if (col < Kc) { xs[2 * (alpha * Kc + col)] = xr; xs[2 * (alpha * Kc + col) + 1] = xi; }
      }
      __syncthreads();
#pragma unroll
      for (int sl = 0; sl < TL_NS; ++sl) {
        if (!live[sl] || tP[sl] != P || tQ[sl] == P) continue;
        double sr = 0, si = 0;  // sum_i conj(L[i][j]) x[i] over this lane's rows i = lq + 4r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double xr = xv[2 * (lq + 4 * r)], xi = xv[2 * (lq + 4 * r) + 1];
          const double lr = accR[sl][r], li = accI[sl][r];
          sr += lr * xr + li * xi;
          si += lr * xi - li * xr;
        }
        sr += __shfl_xor(sr, 16); si += __shfl_xor(si, 16);
        sr += __shfl_xor(sr, 32); si += __shfl_xor(si, 32);
        if (lq == 0) {
          zv[2 * (16 * tQ[sl] + lcol)] -= sr;
          zv[2 * (16 * tQ[sl] + lcol) + 1] -= si;
        }
      }
      __syncthreads();
    }

    write_record(A, xs, sh, mycols, f, n, inst, c, f0, seeds);
  }
}

size_t ls_tile_scratch_stride(int nmax, int Nmax) { return ((size_t)2 * (Nmax + 1) * nmax + 15) & ~(size_t)15; }

// A.scratch / A.scratch_stride / A.work_counter are set by the caller (eaqhm_ls_batch)
int launch_ls_tile(eaqhm_ctx* ctx, LsArgs A, int grid) {
  const int nmax = A.nmax, Kcmax = A.Kcmax;
  const int ldx_max = 16 * TL_NBMAX + 16;
  const int TS = 32;
  const size_t usize_g = (size_t)2 * TS * ldx_max + 3 * TS;
  const size_t usize_c = (size_t)4 * TL_NTMAX * TL_TILE + 2 * TL_WAVES * TL_TILE + 2 * TL_TILE + 2 * 16 * TL_NTMAX + 32;
  const size_t lds_doubles = (usize_c > usize_g ? usize_c : usize_g) + 2 * (size_t)nmax + 4 * (size_t)Kcmax + 16;
  const size_t lds_bytes = lds_doubles * sizeof(double);
  if (lds_bytes > 160 * 1024) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: LDS budget exceeded (tile variant)");
  HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_ls_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
  hipLaunchKernelGGL(eaqhm_ls_tile_kernel, dim3(grid), dim3(TL_THREADS), lds_bytes, ctx->stream, A, TS, ldx_max);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

}  // namespace eaqhm
