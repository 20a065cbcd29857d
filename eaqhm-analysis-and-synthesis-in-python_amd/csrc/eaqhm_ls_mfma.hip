// eaqhm_ls_mfma.hip — the per-frame LS with the Gramian on the FP64 matrix cores (gfx950).
//
// One workgroup of 512 threads (8 waves, 2 per SIMD) owns one frame at a time; frames are pulled from an
// atomic queue (frames differ in size, so static striding leaves tails).
//
//   A1   one thread per active slot: track window, gap fill, running sums, amplitude ratios  -> global
//        scratch Q, r ([sample][slot], slot fastest: coalesced for the next step)        (eaqhm_ls_common.h)
//   A3+B time is cut into chunks of 16 sample PAIRS (u, N-2-u): the pair shares its sincos work because the
//        negative-frequency column at u is the time-reversed positive one (functions.py:284-285).  All
//        threads build the 32 basis rows of a chunk in LDS (planar re/im, [row][column], row stride
//        ≡ 16 (mod 32) doubles so that the four 16-lane groups of an MFMA operand read hit disjoint
//        banks); then every wave runs v_mfma_f64_16x16x4_f64 on its share of the (tile, weight) units:
//            G_p[I][J] += X_I^H diag(w^2 n^p) X_J,  p = 0,1,2,
//        4 real MFMAs per k-step (re·re + im·im, re·im - im·re), accumulators stay in registers for the
//        whole frame.  The signal window is one more basis column, so the right-hand sides are tiles of
//        the same contraction.
//   C    system matrix [[G0,G1],[G1,G2]] + RHS row -> scratch, Cholesky + back substitution
//   D    frequency mismatch, acceptance, record row                                     (eaqhm_ls_common.h)
#include "eaqhm_ls_common.h"
#include "eaqhm_ls_chol.h"

namespace eaqhm {

#define MF_THREADS 512
#define MF_WAVES 8
#define MF_NSLOT 8   // (tile, weight) units per wave per pass: 64 units = 21 complex tiles -> nb <= 6 in one pass

struct MfScratch {
  double* Q;   // (Nmax+1) * nmax
  double* r;   // (Nmax+1) * nmax
  double* T;   // stacked padded system, 16x16 complex tiles (eaqhm_ls_chol.h)
  double* WT;  // inverse diagonal tiles
  double* D0;  // original diagonal of the system (16 per tile row)
};

__host__ __device__ inline size_t mf_tile_doubles(int Kcmax) {
  const size_t nt = 2 * (((size_t)Kcmax + 15) / 16) + 1;
  return nt * (nt + 1) / 2 * 512;
}
__host__ __device__ inline size_t mf_scratch_doubles(int nmax, int Nmax, int Kcmax) {
  const size_t nt = 2 * (((size_t)Kcmax + 15) / 16) + 1;
  return 2 * (size_t)(Nmax + 1) * nmax + mf_tile_doubles(Kcmax) + nt * 2 * TL_TILE + nt * 16;
}

__device__ inline void tile_of(int q, int& I, int& J) {
  I = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
  while ((I + 1) * (I + 2) / 2 <= q) ++I;
  while (I * (I + 1) / 2 > q) --I;
  J = q - I * (I + 1) / 2;
}

extern "C" __global__ void __launch_bounds__(MF_THREADS) eaqhm_ls_mfma_kernel(LsArgs A, int TS, int ldx_max, int min_nb) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (min_nb > 0 && A.cls[LS_BIG_CLASS] == 0) return;   // nothing left over by eaqhm_ls_tile_kernel (uniform across the grid)
  const int tid = threadIdx.x, nt = MF_THREADS;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps the unit bookkeeping in SGPRs
  const int Mmax = 2 * A.Kcmax;
  double* Xre = lds;                         // TS * ldx_max
  double* Xim = Xre + (size_t)TS * ldx_max;  // TS * ldx_max
  double* Wp = Xim + (size_t)TS * ldx_max;   // 3 * TS   weights w^2 n^p per chunk row
  double* rho = Wp + 3 * TS;                 // 2 * nmax
  double* rowj = rho + 2 * A.nmax;           // 2 * Mmax
  double* xs = rowj + 2 * Mmax;              // 2 * Mmax
  double* sh = xs + 2 * Mmax;                // 16
  int* shi = (int*)(sh + 12);
  MfScratch S;
  {
    double* base = A.scratch + (size_t)blockIdx.x * A.scratch_stride;
    S.Q = base;
    S.r = S.Q + (size_t)(A.Nmax + 1) * A.nmax;
    S.T = S.r + (size_t)(A.Nmax + 1) * A.nmax;
    S.WT = S.T + mf_tile_doubles(A.Kcmax);
    S.D0 = S.WT + (2 * (((size_t)A.Kcmax + 15) / 16) + 1) * 2 * TL_TILE;
  }
  const bool seeds = (A.mode == 1) && A.any_seed && (*A.any_seed != 0);
  for (int q = tid; q < 2 * TS * ldx_max; q += nt) Xre[q] = 0.0;  // finite everywhere (rows with weight 0)
  __syncthreads();
  const int PE = TS / 2;  // sample pairs per chunk

  // after eaqhm_ls_tile_kernel (min_nb > 0) only the frames of the last size class are left, usually none
  const int n_items = (min_nb > 0) ? A.cls[LS_BIG_CLASS] : A.n_frames;
  for (;;) {
    if (tid == 0) shi[0] = atomicAdd((min_nb > 0) ? (A.cls + 8 + LS_BIG_CLASS) : A.work_counter, 1);
    __syncthreads();
    const int item = uni(shi[0]);
    __syncthreads();
    if (item >= n_items) break;
    // (wave-uniform, but out of LDS / memory: scalar registers for everything derived from them)
    const int f = uni((min_nb > 0) ? A.cls[16 + (size_t)LS_BIG_CLASS * A.n_frames + item] : item);
    const int c = uni(A.frame_c[f]), wl = uni(A.frame_wl[f]), inst = uni(A.frame_inst[f]);
    const int N = 2 * wl + 1, mid = wl;
    const int n = uni((A.mode == 0) ? A.frame_K[f] : A.ncol[f]);
    const int Kc = 2 * n + 1, C1 = Kc + 1;
    const int nbk = (Kc + 15) >> 4, ntl = 2 * nbk + 1;   // stacked padded tile rows (eaqhm_ls_chol.h)
    const int nb = (C1 + 15) >> 4, C1p = nb << 4;
    const int ldx = C1p + ((nb & 1) ? 0 : 16);  // ≡ 16 (mod 32)
    const int ntiles = nb * (nb + 1) / 2, units = 3 * ntiles;
    const int npass = (units + MF_WAVES * MF_NSLOT - 1) / (MF_WAVES * MF_NSLOT);
    const double f0 = uni((A.mode == 0) ? A.frame_f0[f] : A.f0_stale);
    const int* mycols = (A.mode == 1) ? (A.cols + (size_t)f * A.Kmax) : nullptr;
    const int npairs = mid + 1;  // pairs e = 0..mid: (u, v) = (e-1, N-1-e)

    for (int q = tid; q < 2 * TS * ldx_max; q += nt) Xre[q] = 0.0;  // the factorisation used this region as work space
    if (A.mode == 1) fill_columns(A, S.Q, S.r, rho, mycols, n, N, mid, c, wl, seeds);
    // padding columns of this frame must be zero
    for (int q = tid; q < TS * (C1p - C1); q += nt) {
      int row = q / (C1p - C1), col = C1 + q - row * (C1p - C1);
      Xre[row * ldx + col] = 0.0;
      Xim[row * ldx + col] = 0.0;
    }
    // right-hand-side tile row: zero, its diagonal tile the identity (row 0 is filled from the signal row below)
    for (int q = tid; q < ntl * 512; q += nt) {
      const int Qt = q >> 9, e = q & 511;
      S.T[tile_off(ntl - 1, Qt) + e] = (Qt == ntl - 1 && e < 256 && (e >> 4) == (e & 15)) ? 1.0 : 0.0;
    }
    __syncthreads();

    if (A.mode == 0) {
      // ---- adaptation 0: the stacked padded system straight from the Toeplitz tables (eaqhm_ls_common.h): no basis,
      //      no contraction passes.  Work space: the chunk planes.
      const int TB = A.Kcmax + 1, NCH = 2;
      double* tab = Xre;                                   // [TZ_NQ][TB]
      double* part = tab + (size_t)TZ_NQ * TB;             // [NCH][TZ_NQ][TB]
      double* W2 = part + (size_t)NCH * TZ_NQ * TB;        // [wl+1] each
      const int wpad = ((A.Nmax >> 1) + 8) & ~7;
      double* PA = W2 + wpad;
      double* PB = PA + wpad;
      double* win = PB + wpad;                             // [N]
      double* sig = win + ((A.Nmax + 7) & ~7);             // [N]
      for (int t = tid; t < N; t += nt) {
        win[t] = window_value(1, t, N);
        sig[t] = A.s[(size_t)(c - wl) + t];
      }
      __syncthreads();
      toeplitz_tables(tab, part, W2, PA, PB, sh, win, sig, n, wl, f0 * (2.0 * M_PI / A.fs), tid, TB, NCH);
      const double ssq = sh[0];
      const int ntiles_s = ntl * (ntl + 1) / 2;
      for (int q = tid; q < ntiles_s * 256; q += nt) {
        const int x = q >> 8, e = q & 255, row = e >> 4, col = e & 15;
        int P, Q;
        tile_of(x, P, Q);
        double re = 0.0, im = 0.0;
        if (P == ntl - 1) {                  // right-hand-side tile row: row 0 holds conj(rhs), the rest is padding
          if (Q == ntl - 1) re = (row == col) ? ((row == 0) ? ssq : 1.0) : 0.0;
          else if (row == 0) {
            const int pb = (Q >= nbk) ? 1 : 0, b = 16 * (Q - pb * nbk) + col;
            if (b < Kc) toeplitz_rhs(tab, TB, pb, b, n, re, im);
          }
        } else {
          const int pa = (P >= nbk) ? 1 : 0, pb = (Q >= nbk) ? 1 : 0;
          const int a = 16 * (P - pa * nbk) + row, b = 16 * (Q - pb * nbk) + col;
          if (a < Kc && b < Kc) toeplitz_gram(tab, TB, pa + pb, a, b, n, re, im);
          else if (P == Q && row == col) re = 1.0;          // identity padding inside the two diagonal blocks
        }
        double* t = S.T + tile_off(P, Q) + e;
        t[0] = re; t[256] = im;
      }
    } else
    for (int pass = 0; pass < npass; ++pass) {
      d4 accR[MF_NSLOT], accI[MF_NSLOT];
      int tI[MF_NSLOT], tJ[MF_NSLOT], wsel[MF_NSLOT];
      bool live[MF_NSLOT];
#pragma unroll
      for (int sl = 0; sl < MF_NSLOT; ++sl) {
        accR[sl] = (d4){0, 0, 0, 0};
        accI[sl] = (d4){0, 0, 0, 0};
        const int x = (pass * MF_NSLOT + sl) * MF_WAVES + wave;
        live[sl] = x < units;
        int I = 0, J = 0;
        tile_of(live[sl] ? x / 3 : 0, I, J);
        tI[sl] = I; tJ[sl] = J;
        wsel[sl] = live[sl] ? x % 3 : 0;
      }

      for (int e0 = 0; e0 < npairs; e0 += PE) {
        // ---- build the chunk: rows 2*el (sample u = e-1) and 2*el+1 (sample v = N-1-e)
#pragma clang loop unroll(disable)
        for (int idx = tid; idx < PE * n; idx += nt) {
          const int el = idx / n, j = idx - el * n, e = e0 + el;
          if (e >= npairs) continue;
          const int u = e - 1, v = N - 1 - e;
          double su, cu, sv, cv, ru = 1.0, ru1 = 1.0, rv = 1.0, rv1 = 1.0, pr = 1.0, pi = 0.0;
          if (A.mode == 1) {
            sincos_cw((2.0 * M_PI * S.Q[(size_t)(u + 1) * n + j]) / A.fs, &su, &cu);
            sincos_cw((2.0 * M_PI * S.Q[(size_t)(v + 1) * n + j]) / A.fs, &sv, &cv);
            if (u >= 0) { ru = S.r[(size_t)u * n + j]; rv1 = S.r[(size_t)(v + 1) * n + j]; }
            ru1 = S.r[(size_t)(u + 1) * n + j];
            rv = S.r[(size_t)v * n + j];
            pr = rho[2 * j]; pi = rho[2 * j + 1];
            // positive column at t uses E1(t); negative column at t uses r[mirror+1] * E1(mirror) * rho
            double* xr = Xre + (2 * el) * ldx;
            double* xi = Xim + (2 * el) * ldx;
            if (u >= 0) {
              xr[n + 1 + j] = ru * cu;                  xi[n + 1 + j] = ru * su;
              xr[j] = rv1 * (cv * pr - sv * pi);        xi[j] = rv1 * (cv * pi + sv * pr);
            }
            xr += ldx; xi += ldx;
            xr[n + 1 + j] = rv * cv;                    xi[n + 1 + j] = rv * sv;
            xr[j] = ru1 * (cu * pr - su * pi);          xi[j] = ru1 * (cu * pi + su * pr);
          } else {
            const double fk = (double)(j + 1) * f0;
            double* xr = Xre + (2 * el) * ldx;
            double* xi = Xim + (2 * el) * ldx;
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
        for (int row = tid; row < TS; row += nt) {
          const int e = e0 + (row >> 1);
          const int t = (row & 1) ? (N - 1 - e) : (e - 1);
          double w0 = 0.0, sv = 0.0;
          if (e < npairs && t >= 0) {
            double w = window_value(A.mode == 0, t, N);
            w0 = w * w;
            sv = A.s[(size_t)(c - wl) + t];
          }
          const double nn = (double)(t - mid);
          Wp[row] = w0; Wp[TS + row] = w0 * nn; Wp[2 * TS + row] = w0 * nn * nn;
          Xre[row * ldx + n] = 1.0;  Xim[row * ldx + n] = 0.0;   // DC column
          Xre[row * ldx + Kc] = sv;  Xim[row * ldx + Kc] = 0.0;  // signal column
        }
        __syncthreads();
        // ---- contraction of the chunk
        const int lbase = (lane >> 4) * ldx + (lane & 15);
#pragma unroll
        for (int sl = 0; sl < MF_NSLOT; ++sl) {
          if (!live[sl]) continue;
          const double* wrow = Wp + wsel[sl] * TS + (lane >> 4);
          const double* pAr = Xre + lbase + 16 * tI[sl];
          const double* pAi = Xim + lbase + 16 * tI[sl];
          const double* pBr = Xre + lbase + 16 * tJ[sl];
          const double* pBi = Xim + lbase + 16 * tJ[sl];
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

      // ---- accumulators -> tiles of the stacked padded system [[G0,G1^H],[G1,G2]] + RHS row (eaqhm_ls_chol.h).
      // Base tiles are aligned with the stacked ones; positions beyond Kc inside a block are identity padding.
#pragma unroll
      for (int sl = 0; sl < MF_NSLOT; ++sl) {
        if (!live[sl]) continue;
        const int I = tI[sl], J = tJ[sl], p = wsel[sl];
        const int bl = lane & 15, b = 16 * J + bl;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int al = (lane >> 4) + 4 * rr, a = 16 * I + al;
          const double gr = accR[sl][rr], gi = accI[sl][rr];
          if (a == Kc) {   // signal row: conj(rhs) into row 0 of the RHS tile row, its energy on the diagonal
            if (b < Kc && p == 0) { double* t = S.T + tile_off(ntl - 1, J) + bl; t[0] = gr; t[256] = gi; }
            if (b < Kc && p == 1) { double* t = S.T + tile_off(ntl - 1, nbk + J) + bl; t[0] = gr; t[256] = gi; }
            if (b == Kc && p == 0) { double* t = S.T + tile_off(ntl - 1, ntl - 1); t[0] = gr; t[256] = 0.0; }
          }   // ... and inside the blocks the signal's row / column is one more identity padding position
          if (I >= nbk || J >= nbk) continue;   // the base tile row / column that holds only the signal
          const bool in = (a < Kc) && (b < Kc);
          if (p == 0 || p == 2) {
            const int o = (p == 0) ? 0 : nbk;
            double* t = S.T + tile_off(o + I, o + J) + al * 16 + bl;
            t[0] = in ? gr : ((a == b) ? 1.0 : 0.0);
            t[256] = in ? gi : 0.0;
          } else {   // cross block: G1 is Hermitian, the stacked system needs all of it
            double* t = S.T + tile_off(nbk + I, J) + al * 16 + bl;
            t[0] = in ? gr : 0.0;
            t[256] = in ? gi : 0.0;
            if (I != J) {
              double* u = S.T + tile_off(nbk + J, I) + bl * 16 + al;
              u[0] = in ? gr : 0.0;
              u[256] = in ? -gi : 0.0;
            }
          }
        }
      }
    }
    __syncthreads();

#ifndef EAQHM_EXPERIMENT_NOCHOL   /* (timing experiment: the frame without its factorisation; wrong results) */
    tile_cholesky_memory(S.T, S.WT, S.D0, ntl, Kc, nbk, Xre, xs, A.fault);
#endif
    write_record(A, xs, sh, mycols, f, n, inst, c, f0, seeds);
  }
}

size_t ls_mfma_scratch_stride(int nmax, int Nmax, int Kcmax) {
  return (mf_scratch_doubles(nmax, Nmax, Kcmax) + 15) & ~(size_t)15;
}

// A.scratch / A.scratch_stride / A.work_counter are set by the caller (eaqhm_ls_batch)
int launch_ls_mfma(eaqhm_ctx* ctx, LsArgs A, int grid, int min_nb) {
  const int nmax = A.nmax, Kcmax = A.Kcmax;
  const int nbmax = (Kcmax + 1 + 15) / 16;
  int ldx_max = 16 * nbmax + 16;
  if (ldx_max * 64 < CH_LDS_DOUBLES) ldx_max = (CH_LDS_DOUBLES / 64 + 15) & ~15;   // the factorisation reuses the chunk planes
  // ... and so does the closed-form Gramian of adaptation 0 (tables, partial sums, window, signal)
  const size_t tz_doubles = (size_t)3 * TZ_NQ * (Kcmax + 1) + 3 * (((A.Nmax >> 1) + 8) & ~7) + 2 * ((A.Nmax + 7) & ~7);
  const size_t fixed = (size_t)(2 * nmax + 4 * (2 * Kcmax) + 16) * sizeof(double);
  // rows per chunk: as many as the LDS holds (multiples of 8 = two k-steps; fewer, longer chunks amortise the
  // barriers and the partly filled last round of the basis build)
  int TS = 32;
  while (TS > 8 && (size_t)(2 * TS * ldx_max + 3 * TS) * sizeof(double) + fixed > 158 * 1024) TS -= 8;
  const size_t lds_bytes = (size_t)(2 * TS * ldx_max + 3 * TS) * sizeof(double) + fixed;
  if (lds_bytes > 160 * 1024) return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: problem too large for the MFMA variant");
  if ((size_t)2 * TS * ldx_max < tz_doubles)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: frame size outside the closed-form Gramian's work space");
  if ((size_t)2 * TS * ldx_max < CH_LDS_DOUBLES || 2 * ((Kcmax + 15) / 16) + 1 > CH_NTMAX)
    return ctx->fail(EAQHM_EINVAL, "eaqhm_ls_batch: frame size outside the tile factorisation's work space");
  HIP_TRY(ctx, hipFuncSetAttribute((const void*)eaqhm_ls_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_bytes));
  hipLaunchKernelGGL(eaqhm_ls_mfma_kernel, dim3(grid), dim3(MF_THREADS), lds_bytes, ctx->stream, A, TS, ldx_max, min_nb);
  HIP_TRY(ctx, hipGetLastError());
  return EAQHM_OK;
}

}  // namespace eaqhm
