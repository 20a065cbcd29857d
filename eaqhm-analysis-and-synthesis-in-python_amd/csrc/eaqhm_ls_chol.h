// eaqhm_ls_chol.h — 16x16 complex tile pieces shared by the LS kernels (gfx950, FP64).
#pragma once
#include "eaqhm_ls_common.h"

namespace eaqhm {

typedef double d4 __attribute__((ext_vector_type(4)));

#define TL_LD 17        // tile row stride in LDS (doubles): conflict-free transposing stores
#define TL_TILE (16 * TL_LD)
#define DG_LD 17        // row stride (complex entries) of the diagonal tile and its inverse in LDS: the sixteen rows of
                        // a column start in different banks (a stride of 16 complex = 256 B puts them all in one)
#define DG_TILE (2 * 16 * DG_LD)   // doubles per tile

// ---- diagonal tile: Cholesky factor L and W = L^-1 of a 16x16 Hermitian positive definite tile, by the whole
// workgroup.  The sixteen column steps are a serial chain on the critical path of the frame, so each step is
// made as short as possible: one thread per matrix entry (threads 0-255: D, threads 256-511: the inverse by
// forward elimination on [L | I]), three LDS reads, one reciprocal, one complex multiply-add, one barrier.
//   D  [16][16] complex (interleaved), row-major, lower triangle used;  Z likewise (starts as identity)
//   outputs: Wt planes hold (W^H)[k][j] = conj(W[j][k]) at [k*TL_LD + j];  Ld planes hold L[i][j] at [i*TL_LD + j]
// identity into Z, to be called before the barrier that publishes the diagonal tile in D
__device__ inline void diag_init(double* Z, int tid) {
  const int g = tid >> 8, e = tid & 255, i = e >> 4, k = e & 15;
  if (g == 1) { Z[2 * (i * DG_LD + k)] = (i == k) ? 1.0 : 0.0; Z[2 * (i * DG_LD + k) + 1] = 0.0; }
}

// Singular systems: position q of the tile is a real unknown when q < nvalid (the others are the right-hand-side
// position or identity padding).  dref[q] is the ORIGINAL diagonal entry of the system at that position; a pivot
// (the squared diagonal of L) that is <= PIVOT_TOL of it (eaqhm_ls_common.h) is a breakdown of the factorisation — the
// column is, to rounding, a combination of earlier ones; the reference's inv() raises LinAlgError on the exact form of
// that (functions.py:465, :530).  One thread watches the pivots and counts the frame in *fault (eaqhm_ls_faults).
__device__ inline void diag_coop(double* D, double* Z, double* WtR, double* WtI, double* LdR, double* LdI, int tid,
                                 const double* dref, int nvalid, int* fault) {
  // 2x2 block pivots: seven elimination steps instead of fifteen.  With P = [[p, conj(q)], [q, r]] the pivot block
  // of columns (j, j+1) and a = D[i][j..j+1], b = D[k][j..j+1]:   D[i][k] -= a P^-1 b^H   (i >= k >= j+2), and
  // the rows of the inverse below the block follow the same elimination on [L | I].  Columns / rows inside a
  // block stay raw until the final scaling, which applies the block's own 2x2 Cholesky factor.
  // (Z must hold the identity and D the tile when the workgroup arrives here: diag_init + the caller's barrier)
  // (threads beyond 511 of a larger workgroup only join the barriers)
  const int g = tid >> 8, e = tid & 255, i = e >> 4, k = e & 15;  // entry (i, k) of D (g = 0) or Z (g = 1)
  const int eo = i * DG_LD + k;                                    // ... at its padded position
#pragma clang loop unroll(disable)
  for (int j = 0; j < 14; j += 2) {
    const bool work = (g == 0) ? (k >= j + 2 && i >= k) : ((g == 1) && i >= j + 2 && k <= j + 1);
    if (work) {
      double p = D[2 * (j * DG_LD + j)], r = D[2 * ((j + 1) * DG_LD + j + 1)];
      const double qr = D[2 * ((j + 1) * DG_LD + j)], qi = D[2 * ((j + 1) * DG_LD + j) + 1];
      double det = p * r - (qr * qr + qi * qi);
      if (e == 255 && g == 0) {   // (this thread works in every step)
        if ((j < nvalid && !(p > PIVOT_TOL * dref[j])) || (j + 1 < nvalid && !(det > PIVOT_TOL * dref[j + 1] * p)))
          atomicAdd(fault, 1);
      }
      p = (p > 0.0) ? p : 1.0;
      det = (det > 0.0) ? det : 1.0;   // legitimately only at the RHS position of the last tile (residual ~ 0)
      double dinv = __builtin_amdgcn_rcp(det);
      dinv = dinv * fma(-det, dinv, 2.0);
      dinv = dinv * fma(-det, dinv, 2.0);
      const double a1r = D[2 * (i * DG_LD + j)], a1i = D[2 * (i * DG_LD + j) + 1];
      const double a2r = D[2 * (i * DG_LD + j + 1)], a2i = D[2 * (i * DG_LD + j + 1) + 1];
      // y = P^-1 [x1; x2] * det = [ r x1 - conj(q) x2 ;  -q x1 + p x2 ],  then  out -= (a1 y1 + a2 y2) / det
      double x1r, x1i, x2r, x2i;
      if (g == 0) {   // x = b^H components: conj(D[k][j]), conj(D[k][j+1])
        x1r = D[2 * (k * DG_LD + j)];     x1i = -D[2 * (k * DG_LD + j) + 1];
        x2r = D[2 * (k * DG_LD + j + 1)]; x2i = -D[2 * (k * DG_LD + j + 1) + 1];
      } else {        // x = Z[j][k], Z[j+1][k]
        x1r = Z[2 * (j * DG_LD + k)];       x1i = Z[2 * (j * DG_LD + k) + 1];
        x2r = Z[2 * ((j + 1) * DG_LD + k)]; x2i = Z[2 * ((j + 1) * DG_LD + k) + 1];
      }
      const double y1r = r * x1r - (qr * x2r + qi * x2i), y1i = r * x1i - (qr * x2i - qi * x2r);   // conj(q) x2
      const double y2r = p * x2r - (qr * x1r - qi * x1i), y2i = p * x2i - (qr * x1i + qi * x1r);   // q x1
      const double ur = (a1r * y1r - a1i * y1i) + (a2r * y2r - a2i * y2i);
      const double ui = (a1r * y1i + a1i * y1r) + (a2r * y2i + a2i * y2r);
      double* T = (g == 0) ? D : Z;
      T[2 * eo] -= ur * dinv;
      T[2 * eo + 1] -= ui * dinv;
    }
    __syncthreads();
  }
  // final scaling with each pivot block's own Cholesky factor [[l11, 0], [l21, l22]]
  if (g < 2) {
    const int pj = ((g == 0) ? k : i) & ~1;    // first column (D) / row (Z) of the entry's pivot block
    double p = D[2 * (pj * DG_LD + pj)], r = D[2 * ((pj + 1) * DG_LD + pj + 1)];
    const double qr = D[2 * ((pj + 1) * DG_LD + pj)], qi = D[2 * ((pj + 1) * DG_LD + pj) + 1];
    if (e == 255 && g == 0) {   // the last pivot block (14, 15) is not covered by the elimination steps
      const double s = r - (qr * qr + qi * qi) / p;
      if ((14 < nvalid && !(p > PIVOT_TOL * dref[14])) || (15 < nvalid && !(s > PIVOT_TOL * dref[15]))) atomicAdd(fault, 1);
    }
    p = (p > 0.0) ? p : 1.0;
    double i11 = __builtin_amdgcn_rsq(p);
    i11 = i11 * fma(-0.5 * p * i11, i11, 1.5);
    i11 = i11 * fma(-0.5 * p * i11, i11, 1.5);
    const double l21r = qr * i11, l21i = qi * i11;
    double s22 = r - (l21r * l21r + l21i * l21i);
    s22 = (s22 > 0.0) ? s22 : 1.0;
    double i22 = __builtin_amdgcn_rsq(s22);
    i22 = i22 * fma(-0.5 * s22 * i22, i22, 1.5);
    i22 = i22 * fma(-0.5 * s22 * i22, i22, 1.5);
    if (g == 0) {
      double lr, li;
      const bool second = (k & 1) != 0;
      if (i < k) { lr = 0.0; li = 0.0; }
      else if (!second) {                       // column pj
        if (i == pj) { lr = p * i11; li = 0.0; }
        else if (i == pj + 1) { lr = l21r; li = l21i; }
        else { lr = D[2 * eo] * i11; li = D[2 * eo + 1] * i11; }
      } else {                                  // column pj+1
        if (i == pj + 1) { lr = s22 * i22; li = 0.0; }
        else {   // (a2 - (a1 / l11) conj(l21)) / l22
          const double a1r = D[2 * (i * DG_LD + pj)] * i11, a1i = D[2 * (i * DG_LD + pj) + 1] * i11;
          lr = (D[2 * eo] - (a1r * l21r + a1i * l21i)) * i22;
          li = (D[2 * eo + 1] - (a1i * l21r - a1r * l21i)) * i22;
        }
      }
      LdR[i * TL_LD + k] = lr;
      LdI[i * TL_LD + k] = li;
    } else {   // W rows of the block: W[pj] = Z[pj] / l11,  W[pj+1] = (Z[pj+1] - l21 W[pj]) / l22;  stored as W^H
      double wr = 0.0, wi = 0.0;
      if (k <= i) {
        const double z1r = Z[2 * (pj * DG_LD + k)] * i11, z1i = Z[2 * (pj * DG_LD + k) + 1] * i11;
        if ((i & 1) == 0) { wr = z1r; wi = z1i; }
        else {
          wr = (Z[2 * eo] - (l21r * z1r - l21i * z1i)) * i22;
          wi = (Z[2 * eo + 1] - (l21r * z1i + l21i * z1r)) * i22;
        }
      }
      WtR[k * TL_LD + i] = wr;
      WtI[k * TL_LD + i] = -wi;
    }
  }
  __syncthreads();
}


// ---- diagonal tile on the matrix cores, two waves, no workgroup barrier ----------------------------------------
// The same 2x2-block-pivot elimination as diag_coop, but the tile (and the inverse in the making) stays in MFMA
// accumulator layout — lane (lq, lcol), register r <-> entry (lq + 4 r, lcol) — and every elimination step is a
// complex rank-2 update, i.e. a real 16x16x4 product: ONE v_mfma_f64_16x16x4_f64 per real / imaginary plane,
//     T[i][k] -= a1[i] c1[k] + a2[i] c2[k],   a_s = columns j, j+1 of the tile,  c = P^-1 [conj a1[k]; conj a2[k]],
//     Z[i][k] -= a1[i] d1[k] + a2[i] d2[k],   d = P^-1 [Z[j][k]; Z[j+1][k]]          (forward elimination on [L | I]),
// with the operand (row i, k-index kk) = (a1R, a1I, a2R, a2I)[kk] and the products' signs folded into the second
// operand.  Rows / columns that are finished (i, k < j+2) are masked out of the operands.
// The work is a chain of dependent scalar-ish steps, so it is split over two waves that run as a pipeline:
//   diag_D  (the wave that owns the tile)  eliminates the tile itself and POSTS, per step, the two pivot columns and
//           the pivot block to LDS, then raises a step counter (release);
//   diag_Z  (a helper wave)  follows one step behind (acquire), applies the same eliminations to the identity and
//           finishes rows j, j+1 of W = L^-1 with the block's own Cholesky factor.
// Neither touches a workgroup barrier, so the other waves run their trailing updates meanwhile (look-ahead).
// The tile is Hermitian and BOTH triangles are kept up to date, so row j of the tile, T[j][k] = conj(a1[k]), is at
// once the conjugated column the second operand is made of and (conjugated back) the first operand; a row sits in
// sixteen lanes of one register, and ds_bpermute hands it to every lane without a round trip through LDS memory.
//   post     LDS: [8 steps][2 rows][16][re, im]  rows j, j+1 of the tile at step j/2 (they contain the pivot block)
//   Wt*      (W^H)[k][j] = conj(W[j][k]) at [k*TL_LD + j];     Ld* (want_L): L[i][j] at [i*TL_LD + j], rows > j+1 only
//   dref / nvalid / fault: as for diag_coop
#define DGP_DOUBLES 512
__device__ inline double rdlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double bperm(double v, int byte_addr) {
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}
__device__ inline double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// Bounded wait on the pipeline's step counter: every wave reaches an exit even if the protocol were broken; the frame
// is then counted in fault[1] (a counter of its own: a stalled hand-shake is a bug of this library, not a singular
// matrix) and the host raises RuntimeError.
__device__ inline bool spin_until(int* flag, int target) {
  for (int spins = 0; spins < (1 << 16); ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

//   postB    LDS: [7 steps][64 lanes][re-plane operand, im-plane operand]  the second MFMA operands of each elimination
//            step, as this wave uses them itself: panel_follow applies the same steps to the panel tiles of the column
#define DGB_DOUBLES (7 * 64 * 2)
__device__ __attribute__((always_inline)) inline void diag_D(d4 R, d4 I, double* post, double* postB, int* flag,
                                                              int flag_base, double* dump, double* LdR, double* LdI,
                                                              bool want_L) {
  const int lane = threadIdx.x & 63, lq = lane >> 4, lcol = lane & 15;
  const bool hi = lq >= 2, odd = (lq & 1) != 0;   // which of (c1, c2) / (re, im) this lane's operand entry is
  const double sg = hi ? 1.0 : -1.0;
#pragma unroll
  for (int j = 0; j < 16; j += 2) {
    const int j1 = j + 1, st = j >> 1, rg = j >> 2;   // rows j and j+1 sit in the same register
    // ---- pivot block (wave-uniform, scalar registers)
    double p = rdlane(R[rg], (j & 3) * 16 + j);
    const double r = rdlane(R[rg], (j1 & 3) * 16 + j1);
    const double qr = rdlane(R[rg], (j1 & 3) * 16 + j), qi = rdlane(I[rg], (j1 & 3) * 16 + j);
    // ---- rows j, j+1 to every lane: X = the row this lane's operand entry belongs to, Y = the other one
    const int aX = (((hi ? j1 : j) & 3) * 16 + lcol) * 4, aY = (((hi ? j : j1) & 3) * 16 + lcol) * 4;
    const double Xr = bperm(R[rg], aX), Xi = bperm(I[rg], aX), Yr = bperm(R[rg], aY), Yi = bperm(I[rg], aY);
    // ---- post the two rows for diag_Z (the lanes that hold them; the others write to their dump slot)
    {
      const bool first = lq == (j & 3), hold = first || (lq == (j1 & 3));
      double* dst = hold ? (post + st * 64 + (first ? 0 : 32) + lcol * 2) : (dump + lane * 2);
      dst[0] = R[rg];
      dst[1] = I[rg];
    }
    double det = p * r - (qr * qr + qi * qi);
    p = (p > 0.0) ? p : 1.0;
    det = (det > 0.0) ? det : 1.0;   // legitimately only at the RHS position of the last tile (residual ~ 0)
    double dinv = __builtin_amdgcn_rcp(det);
    dinv = dinv * fma(-det, dinv, 2.0);
    dinv = dinv * fma(-det, dinv, 2.0);
    if (j < 14) {
      // P^-1 [x1; x2] = ( [ r x1 - conj(q) x2 ;  p x2 - q x1 ] ) / det,  x_t[k] = conj(a_t[k]) = T[j+t-1][k].
      // A lane needs ONE real entry of one of the two:  res = u X - v Y,  (u, X, v, Y) = (r, x1, conj q, x2) for c1
      // (lq < 2), (p, x2, q, x1) for c2 (lq >= 2);  second operand of the real-plane product: odd ? Im res : -Re res,
      // of the imaginary plane: odd ? -Re res : -Im res.  Finished columns (k < j+2) get a zero operand; the first
      // operand (row i = lcol) is a_X[i] = conj(X): (Re, -Im) by `odd`, zero for finished rows.
      const double m = (lcol >= j + 2) ? dinv : 0.0;
      const double u = (hi ? p : r) * m, vr = qr * m, vi = (qi * sg) * m;
      double asel = odd ? -Xi : Xr;
      asel = (lcol >= j + 2) ? asel : 0.0;
      const double rr_ = fma(u, Xr, -fma(vr, Yr, -(vi * Yi))), ri_ = fma(u, Xi, -fma(vr, Yi, vi * Yr));
      const double bre = odd ? ri_ : -rr_, bim = odd ? -rr_ : -ri_;
      if (postB) { postB[(st * 64 + lane) * 2] = bre; postB[(st * 64 + lane) * 2 + 1] = bim; }
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, bre, R, 0, 0, 0);
      I = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, bim, I, 0, 0, 0);
    }
    __hip_atomic_store(flag, flag_base + st + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (want_L) {   // (last tile of a frame only; only rows beyond the block are ever read: the RHS row)
      const double a1r = hi ? Yr : Xr, a1i = hi ? -Yi : -Xi, a2r = hi ? Xr : Yr, a2i = hi ? -Xi : -Yi;   // row i = lcol
      const double i11 = rsqrt_nr(p);
      const double l21r = qr * i11, l21i = qi * i11;
      double s22 = r - (l21r * l21r + l21i * l21i);
      s22 = (s22 > 0.0) ? s22 : 1.0;
      const double i22 = rsqrt_nr(s22);
      const double l1r = a1r * i11, l1i = a1i * i11;
      const double l2r = (a2r - (l1r * l21r + l1i * l21i)) * i22, l2i = (a2i - (l1i * l21r - l1r * l21i)) * i22;
      LdR[lcol * TL_LD + j] = l1r;   LdI[lcol * TL_LD + j] = l1i;
      LdR[lcol * TL_LD + j1] = l2r;  LdI[lcol * TL_LD + j1] = l2i;
    }
  }
}

// zs: 64 doubles of wave-private LDS (per-row factors)
__device__ __attribute__((always_inline)) inline void diag_Z(const double* post, int* flag, int flag_base, double* zs,
                                                              double* WtR, double* WtI, const double* dref, int nvalid,
                                                              int* fault) {
  const int lane = threadIdx.x & 63, lq = lane >> 4, lcol = lane & 15;
  const bool hi = lq >= 2, odd = (lq & 1) != 0;
  const double sg = hi ? 1.0 : -1.0;
  const int offA = (hi ? 32 : 0) + lcol * 2 + (odd ? 1 : 0);   // this lane's first-operand entry inside a step's rows
  double* facrow = zs;   // [16][4]  per row of W: {iota, lambda re, lambda im, kappa}
  int stuck = 0;
  d4 ZR, ZI = (d4){0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) ZR[r] = (lq + 4 * r == lcol) ? 1.0 : 0.0;
#pragma unroll
  for (int j = 0; j < 14; j += 2) {
    const int j1 = j + 1, st = j >> 1, rg = j >> 2;
    // rows j, j+1 of Z (final after the previous step) to every lane
    const int aX = (((hi ? j1 : j) & 3) * 16 + lcol) * 4, aY = (((hi ? j : j1) & 3) * 16 + lcol) * 4;
    const double Xr = bperm(ZR[rg], aX), Xi = bperm(ZI[rg], aX), Yr = bperm(ZR[rg], aY), Yi = bperm(ZI[rg], aY);
    if (!spin_until(flag, flag_base + st + 1)) stuck = 1;
    const double* rows = post + st * 64;   // T[j][k] at [2k], T[j+1][k] at [32 + 2k]
    double p = rows[2 * j];
    const double qr = rows[2 * j1], qi = -rows[2 * j1 + 1], r = rows[32 + 2 * j1];   // q = T[j+1][j] = conj(T[j][j+1])
    double det = p * r - (qr * qr + qi * qi);
    p = (p > 0.0) ? p : 1.0;
    det = (det > 0.0) ? det : 1.0;
    double dinv = __builtin_amdgcn_rcp(det);
    dinv = dinv * fma(-det, dinv, 2.0);
    dinv = dinv * fma(-det, dinv, 2.0);
    double asel = rows[offA];            // a_X[i] = conj(T[row][i]):  (Re, -Im) by `odd`
    asel = odd ? -asel : asel;
    asel = (lcol >= j + 2) ? asel : 0.0;
    const double u = (hi ? p : r) * dinv, vr = qr * dinv, vi = (qi * sg) * dinv;
    const double rr_ = fma(u, Xr, -fma(vr, Yr, -(vi * Yi))), ri_ = fma(u, Xi, -fma(vr, Yi, vi * Yr));
    const double zre = odd ? ri_ : -rr_, zim = odd ? -rr_ : -ri_;
    ZR = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, zre, ZR, 0, 0, 0);
    ZI = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, zim, ZI, 0, 0, 0);
  }
  // ---- every pivot block's own Cholesky factor [[l11, 0], [l21, l22]] at once (lane b <-> block b), the collapsed-
  //      pivot check, and the rows of W:  W[j] = Z[j] / l11,  W[j+1] = (Z[j+1] - l21 W[j]) / l22
  if (!spin_until(flag, flag_base + 8)) stuck = 1;
  if (stuck && lane == 0) atomicAdd(fault + 1, 1);
  {
    const int b = lane & 7, j = 2 * b;
    const double* rows = post + b * 64;
    double p = rows[2 * j];
    const double qr = rows[2 * (j + 1)], qi = -rows[2 * (j + 1) + 1], r = rows[32 + 2 * (j + 1)];
    const double det = p * r - (qr * qr + qi * qi);
    const bool bad = (j < nvalid && !(p > PIVOT_TOL * dref[j])) || (j + 1 < nvalid && !(det > PIVOT_TOL * dref[j + 1] * p));
    p = (p > 0.0) ? p : 1.0;
    const double i11 = rsqrt_nr(p);
    const double l21r = qr * i11, l21i = qi * i11;
    double s22 = r - (l21r * l21r + l21i * l21i);
    s22 = (s22 > 0.0) ? s22 : 1.0;
    const double i22 = rsqrt_nr(s22);
    // W_row = (Z_row - lambda (kappa Z_partner)) iota:  first row of a block {i11, 0, 0, 0}, second {i22, l21, i11}
    facrow[j * 4 + 0] = i11; facrow[j * 4 + 1] = 0.0; facrow[j * 4 + 2] = 0.0; facrow[j * 4 + 3] = 0.0;
    facrow[j * 4 + 4] = i22; facrow[j * 4 + 5] = l21r; facrow[j * 4 + 6] = l21i; facrow[j * 4 + 7] = i11;
    if (__ballot(bad) != 0ull && lane == 0) atomicAdd(fault, 1);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    const double io = facrow[row * 4], lr = facrow[row * 4 + 1], li = facrow[row * 4 + 2], ka = facrow[row * 4 + 3];
    // partner row (row ^ 1) of the same column sits in lane ^ 16, same register
    const double pr = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(ZR[r]), 0x401F),
                                       __builtin_amdgcn_ds_swizzle(__double2loint(ZR[r]), 0x401F));
    const double pi = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(ZI[r]), 0x401F),
                                       __builtin_amdgcn_ds_swizzle(__double2loint(ZI[r]), 0x401F));
    const double wpr = ka * pr, wpi = ka * pi;
    const double wr = (ZR[r] - (lr * wpr - li * wpi)) * io, wi = (ZI[r] - (lr * wpi + li * wpr)) * io;
    WtR[lcol * TL_LD + row] = wr;
    WtI[lcol * TL_LD + row] = -wi;
  }
}


// ---- panel tiles without the inverse: a third role in the pipeline ------------------------------------------------
// A panel tile A = T[P][jb] (P > jb) becomes L[P][jb] = A L_jj^-H.  Instead of waiting for W = L_jj^-1 and multiplying
// (two workgroup barriers and the tail of diag_Z on the critical path of every tile row), the tile's owner APPLIES THE
// ELIMINATION STEPS of the diagonal tile to it as diag_D posts them: step (j, j+1) is
//     A[i][k] -= A[i][j] c1[k] + A[i][j+1] c2[k],   k >= j+2,
// with exactly the (c1, c2) operand diag_D uses on its own tile (postB); the first operand is the tile's own column pair
// (j, j+1), handed to the MFMA operand lanes through 64 doubles of wave-private LDS.  What is left are the raw columns
// of each pivot block, finished with the block's 2x2 Cholesky factor [[l11, 0], [l21, l22]]:
//     L[:, j] = A[:, j] / l11,   L[:, j+1] = (A[:, j+1] - L[:, j] conj(l21)) / l22        (lane ^ 1 holds the partner column).
// The wave may arrive late (after its trailing updates): the posts of all steps stay in LDS until the stage ends.
//   tab  64 doubles, fac 64 doubles of wave-private LDS
__device__ inline void panel_block_factors(const double* post, double* fac) {   // lane b <-> pivot block b
  const int lane = threadIdx.x & 63, b = lane & 7, j = 2 * b;
  const double* rows = post + b * 64;
  double p = rows[2 * j];
  const double qr = rows[2 * (j + 1)], qi = -rows[2 * (j + 1) + 1], r = rows[32 + 2 * (j + 1)];   // q = T[j+1][j]
  p = (p > 0.0) ? p : 1.0;
  const double i11 = rsqrt_nr(p);
  const double l21r = qr * i11, l21i = qi * i11;
  double s22 = r - (l21r * l21r + l21i * l21i);
  s22 = (s22 > 0.0) ? s22 : 1.0;
  const double i22 = rsqrt_nr(s22);
  // per column of the tile: {iota, lambda re, lambda im, kappa};  first column of a block {i11, 0, 0, 0}, second {i22, l21, i11}
  fac[j * 4 + 0] = i11; fac[j * 4 + 1] = 0.0; fac[j * 4 + 2] = 0.0; fac[j * 4 + 3] = 0.0;
  fac[j * 4 + 4] = i22; fac[j * 4 + 5] = l21r; fac[j * 4 + 6] = l21i; fac[j * 4 + 7] = i11;
  __builtin_amdgcn_wave_barrier();
}
__device__ __attribute__((always_inline)) inline bool panel_follow(d4& R, d4& I, const double* postB, int* flag,
                                                                    int flag_base, double* tab) {
  const int lane = threadIdx.x & 63, lq = lane >> 4, lcol = lane & 15;
  bool ok = true;
#pragma clang loop unroll(disable)
  for (int st = 0; st < 7; ++st) {
    const int j = 2 * st;
    // the tile's columns j, j+1 (lanes lcol = j, j+1; rows lq + 4 r) -> tab[row][column][re, im]
    if ((lcol >> 1) == st) {
      const int cb = (lcol & 1) * 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tab[(lq + 4 * r) * 4 + cb] = R[r];
        tab[(lq + 4 * r) * 4 + cb + 1] = I[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // first operand: row i = lcol, k-slot lq = (a1 re, a1 im, a2 re, a2 im)
    const double asel = tab[lcol * 4 + lq];
    if (!spin_until(flag, flag_base + st + 1)) ok = false;
    const double bre = postB[(st * 64 + lane) * 2], bim = postB[(st * 64 + lane) * 2 + 1];
    R = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, bre, R, 0, 0, 0);
    I = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, bim, I, 0, 0, 0);
    __builtin_amdgcn_wave_barrier();   // (tab is rewritten by the next step)
    (void)j;
  }
  return ok;
}
// finish the columns of a followed tile with the pivot blocks' own factors (panel_block_factors)
__device__ __attribute__((always_inline)) inline void panel_finish(d4& R, d4& I, const double* fac) {
  const int lcol = threadIdx.x & 15;
  const double io = fac[lcol * 4], lr = fac[lcol * 4 + 1], li = fac[lcol * 4 + 2], ka = fac[lcol * 4 + 3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // partner column (lcol ^ 1) of the same row sits in lane ^ 1, same register
    const double pr = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(R[r]), 0x041F),
                                       __builtin_amdgcn_ds_swizzle(__double2loint(R[r]), 0x041F));
    const double pi = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(I[r]), 0x041F),
                                       __builtin_amdgcn_ds_swizzle(__double2loint(I[r]), 0x041F));
    const double wpr = ka * pr, wpi = ka * pi;                  // L[:, j] of the block (zero for a first column)
    const double xr = (R[r] - (wpr * lr + wpi * li)) * io, xi = (I[r] - (wpi * lr - wpr * li)) * io;
    R[r] = xr; I[r] = xi;
  }
}


// ---- the same two-wave pipeline for a REAL symmetric positive definite tile (adaptation 0: the even / odd systems
//      of a0_frame).  Rank-2 update of a real tile = one 16x16x4 product with two of its four k-slots used:
//      A[i][kk] = (a1[i], a2[i], 0, 0),  B[kk][k] = -(c1[k], c2[k], 0, 0),  c = P^-1 [a1[k]; a2[k]].
//   post: [8 steps][2 rows][16]   rows j, j+1 of the tile (symmetric: row j = column j)
__device__ __attribute__((always_inline)) inline void diag_Dr(d4 R, double* post, int* flag, int flag_base, double* dump,
                                                               double* Ld, bool want_L) {
  const int lane = threadIdx.x & 63, lq = lane >> 4, lcol = lane & 15;
  const bool odd = (lq & 1) != 0, used = lq < 2;
#pragma unroll
  for (int j = 0; j < 16; j += 2) {
    const int j1 = j + 1, st = j >> 1, rg = j >> 2;
    double p = rdlane(R[rg], (j & 3) * 16 + j);
    const double r = rdlane(R[rg], (j1 & 3) * 16 + j1), q = rdlane(R[rg], (j1 & 3) * 16 + j);
    const int aX = (((odd ? j1 : j) & 3) * 16 + lcol) * 4, aY = (((odd ? j : j1) & 3) * 16 + lcol) * 4;
    const double X = bperm(R[rg], aX), Y = bperm(R[rg], aY);
    {
      const bool first = lq == (j & 3), hold = first || (lq == (j1 & 3));
      double* dst = hold ? (post + st * 32 + (first ? 0 : 16) + lcol) : (dump + lane);
      dst[0] = R[rg];
    }
    double det = p * r - q * q;
    p = (p > 0.0) ? p : 1.0;
    det = (det > 0.0) ? det : 1.0;   // legitimately only at the RHS position of the last tile (residual ~ 0)
    double dinv = __builtin_amdgcn_rcp(det);
    dinv = dinv * fma(-det, dinv, 2.0);
    dinv = dinv * fma(-det, dinv, 2.0);
    if (j < 14) {
      // c1 = (r x1 - q x2) / det (k-slot 0),  c2 = (p x2 - q x1) / det (k-slot 1);  X is this lane's own row
      const bool on = used && (lcol >= j + 2);
      const double m = on ? dinv : 0.0;
      const double res = ((odd ? p : r) * X - q * Y) * m;
      const double asel = on ? X : 0.0;
      R = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, -res, R, 0, 0, 0);
    }
    __hip_atomic_store(flag, flag_base + st + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (want_L) {   // (last tile of a system only; only rows beyond the block are ever read: the RHS row)
      const double a1 = odd ? Y : X, a2 = odd ? X : Y;   // row i = lcol of columns j, j+1
      const double i11 = rsqrt_nr(p), l21 = q * i11;
      double s22 = r - l21 * l21;
      s22 = (s22 > 0.0) ? s22 : 1.0;
      const double i22 = rsqrt_nr(s22);
      const double l1 = a1 * i11, l2 = (a2 - l1 * l21) * i22;
      Ld[lcol * TL_LD + j] = l1;
      Ld[lcol * TL_LD + j1] = l2;
    }
  }
}

// zs: 64 doubles of wave-private LDS (per-row factors)
__device__ __attribute__((always_inline)) inline void diag_Zr(const double* post, int* flag, int flag_base, double* zs,
                                                               double* Wt, const double* dref, int nvalid, int* fault) {
  const int lane = threadIdx.x & 63, lq = lane >> 4, lcol = lane & 15;
  const bool odd = (lq & 1) != 0, used = lq < 2;
  double* facrow = zs;   // [16][4]  per row of W: {iota, lambda, kappa, -}
  int stuck = 0;
  d4 Z;
#pragma unroll
  for (int r = 0; r < 4; ++r) Z[r] = (lq + 4 * r == lcol) ? 1.0 : 0.0;
#pragma unroll
  for (int j = 0; j < 14; j += 2) {
    const int j1 = j + 1, st = j >> 1, rg = j >> 2;
    const int aX = (((odd ? j1 : j) & 3) * 16 + lcol) * 4, aY = (((odd ? j : j1) & 3) * 16 + lcol) * 4;
    const double X = bperm(Z[rg], aX), Y = bperm(Z[rg], aY);
    if (!spin_until(flag, flag_base + st + 1)) stuck = 1;
    const double* rows = post + st * 32;   // T[j][k] at [k], T[j+1][k] at [16 + k]
    double p = rows[j];
    const double q = rows[16 + j], r = rows[16 + j1];
    double det = p * r - q * q;
    p = (p > 0.0) ? p : 1.0;
    det = (det > 0.0) ? det : 1.0;
    double dinv = __builtin_amdgcn_rcp(det);
    dinv = dinv * fma(-det, dinv, 2.0);
    dinv = dinv * fma(-det, dinv, 2.0);
    const bool on = used && (lcol >= j + 2);
    double asel = rows[(odd ? 16 : 0) + lcol];   // a_s[i] = T[j+s-1][i]
    asel = on ? asel : 0.0;
    const double res = used ? ((odd ? p : r) * X - q * Y) * dinv : 0.0;
    Z = __builtin_amdgcn_mfma_f64_16x16x4f64(asel, -res, Z, 0, 0, 0);
  }
  if (!spin_until(flag, flag_base + 8)) stuck = 1;
  if (stuck && lane == 0) atomicAdd(fault + 1, 1);
  {
    const int b = lane & 7, j = 2 * b;
    const double* rows = post + b * 32;
    double p = rows[j];
    const double q = rows[16 + j], r = rows[16 + j + 1];
    const double det = p * r - q * q;
    const bool bad = (j < nvalid && !(p > PIVOT_TOL * dref[j])) || (j + 1 < nvalid && !(det > PIVOT_TOL * dref[j + 1] * p));
    p = (p > 0.0) ? p : 1.0;
    const double i11 = rsqrt_nr(p), l21 = q * i11;
    double s22 = r - l21 * l21;
    s22 = (s22 > 0.0) ? s22 : 1.0;
    const double i22 = rsqrt_nr(s22);
    // W_row = (Z_row - lambda (kappa Z_partner)) iota:  first row of a block {i11, 0, 0}, second {i22, l21, i11}
    facrow[j * 4 + 0] = i11; facrow[j * 4 + 1] = 0.0; facrow[j * 4 + 2] = 0.0;
    facrow[j * 4 + 4] = i22; facrow[j * 4 + 5] = l21; facrow[j * 4 + 6] = i11;
    if (__ballot(bad) != 0ull && lane == 0) atomicAdd(fault, 1);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = lq + 4 * r;
    const double io = facrow[row * 4], la = facrow[row * 4 + 1], ka = facrow[row * 4 + 2];
    const double pr = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(Z[r]), 0x401F),
                                       __builtin_amdgcn_ds_swizzle(__double2loint(Z[r]), 0x401F));   // row ^ 1
    Wt[lcol * TL_LD + row] = (Z[r] - la * (ka * pr)) * io;
  }
}


// ------------------------------------------------------------------------------------------------------------
// Tile Cholesky through memory, for systems too large for the register file (eaqhm_ls_mfma_kernel).
//
// Stacked, padded system of order 16*nt, nt = 2*nbk + 1, nbk = ceil(Kc/16): tile rows [0, nbk) = amplitude block,
// [nbk, 2 nbk) = slope block (each padded to 16*nbk with identity rows/columns), tile row 2*nbk = right-hand-side
// row (row 0; rows 1..15 identity padding).  Only tiles P >= Q are stored: tile (P,Q) at T + tile_off(P,Q), real
// plane [256] then imaginary plane [256].  Before its column is factorised a tile holds the Hermitian block
// row-major [row][col] (the MFMA accumulator order: lane l, register r <-> index 64 r + l); afterwards it holds
// L[P][Q] k-major [col][row], the order in which the MFMA operand loads of later columns are coalesced.
//
// Left-looking by BLOCKS of CH_W tile columns.  The factor tiles of a 48 kHz frame (2-3 MB per workgroup, 0.8 GB for the
// grid) live in HBM, not in L2; a column-by-column left-looking update reads every tile L[P][j] once per later column —
// nt^3/6 tile reads, 20 MB per frame, 3-6 TB per launch measured, three times what HBM delivers while the MFMAs of
// the update would need it.  So the update is applied to CH_W columns at once:
//   (1) block update   T[P][Q0+c] -= sum_{j<Q0} L[P][j] L[Q0+c][j]^H  for the CH_W columns of the block and all rows
//       P >= Q0: a wave takes CH_RB rows at a time, loads each A tile L[P][j] ONCE for the columns of the block (the B
//       tiles of a j are shared by all waves: L1/L2), accumulates in registers and writes the tiles back in place;
//   (2) the columns of the block one after another, as before, with the j-loop running over the block's own earlier
//       columns only: the diagonal tile is factorised and inverted in LDS by the whole workgroup (diag_coop) and the
//       panel tiles are multiplied by the inverse (MFMA) and stored k-major.
// The right-hand side rides along as the last tile row, so the forward substitution is free; the back
// substitution walks the tile columns from the stored factor.
#define CH_MB 6          // tiles per wave, column and register group (single-column part)
#define CH_W 3           // tile columns per block
#define CH_RB 2          // tile rows per wave and register group of the block update
// (measured on the 48 kHz workload, ms per adaptation>=1 launch: 2 x 3 tiles with three real products per complex one
//  576; 1 x 4: 581; 1 x 5: 578; 3 x 2: 584; 2 x 4 with four products — two accumulators per tile — 593; 2 x 4 with three: 622, spills)
#define CH_NTMAX 96      // tile rows the work space is sized for (system order 16 * 96)
__device__ inline size_t tile_off(int P, int Q) { return ((size_t)P * (P + 1) / 2 + Q) * 512; }
#define CH_LDS_DOUBLES (2 * DG_TILE + 4 * TL_TILE + 16 * 2 * TL_TILE + 2 * 16 * CH_NTMAX + 32 + 16)   // (up to 16 waves)

// One register group of the block update (phase (1) of tile_cholesky_memory): rows Pg (and Pg + 8 when RB = 2), the CH_W
// columns of the block, T[P][Q0+c] -= sum_{j<Q0} L[P][j] L[Q0+c][j]^H.  The K-loop is branch-free: every one of the
// RB x CH_W tiles is accumulated (a tile above the diagonal or beyond the last column costs its MFMAs — three tiles
// per block at most — and is simply not written back), the sums of the three-product form are formed before the MFMAs so
// that those issue back to back, and two k-steps per trip ping-pong between two operand sets (no register copies).
template <int RB, int RS>   // RB rows Pg, Pg + RS
__device__ __attribute__((always_inline)) inline void chol_block_unit(double* __restrict__ T, int Q0, int Wc, int Pg, int lq, int lcol) {
  d4 cR[RB][CH_W], cI[RB][CH_W], c3[RB][CH_W];
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int c = 0; c < CH_W; ++c) {
      cR[m][c] = (d4){0, 0, 0, 0}; cI[m][c] = (d4){0, 0, 0, 0}; c3[m][c] = (d4){0, 0, 0, 0};
    }
  const double* Bp[CH_W];
  const double* Ap[RB];
#pragma unroll
  for (int c = 0; c < CH_W; ++c) Bp[c] = T + tile_off((c < Wc) ? (Q0 + c) : Q0, 0) + lq * 16 + lcol;
#pragma unroll
  for (int m = 0; m < RB; ++m) Ap[m] = T + tile_off(Pg + RS * m, 0) + lq * 16 + lcol;
  // operand sets x (current k-step) and y (next): tile j = it/4 of a row starts 512 j doubles after tile 0, k-step
  // ks = it%4 is 64 doubles further; the requests of step it+1 are issued before the MFMAs of step it
  double xbR[CH_W], xbI[CH_W], xaR[RB], xaI[RB], ybR[CH_W], ybI[CH_W], yaR[RB], yaI[RB];
#pragma unroll
  for (int c = 0; c < CH_W; ++c) { xbR[c] = Bp[c][0]; xbI[c] = Bp[c][256]; }
#pragma unroll
  for (int m = 0; m < RB; ++m) { xaR[m] = Ap[m][0]; xaI[m] = Ap[m][256]; }
  const int nit = 4 * Q0;   // (even)
#define CH_BU_STEP(aR, aI, bR, bI, naR, naI, nbR, nbI, nx)                                                   \
  {                                                                                                           \
    const size_t off = (size_t)((nx) >> 2) * 512 + (size_t)((nx) & 3) * 64;                                   \
    _Pragma("unroll") for (int c = 0; c < CH_W; ++c) { nbR[c] = Bp[c][off]; nbI[c] = Bp[c][off + 256]; }      \
    _Pragma("unroll") for (int m = 0; m < RB; ++m) { naR[m] = Ap[m][off]; naI[m] = Ap[m][off + 256]; }        \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    double sA[RB], dB[CH_W];                                                                                  \
    _Pragma("unroll") for (int m = 0; m < RB; ++m) sA[m] = aR[m] + aI[m];                                     \
    _Pragma("unroll") for (int c = 0; c < CH_W; ++c) dB[c] = bI[c] - bR[c];                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    /* P1 = re re, P2 = im im, P3 = (re + im)(im' - re'):  Re = P1 + P2,  -Im = P3 + P1 - P2 */               \
    _Pragma("unroll") for (int m = 0; m < RB; ++m)                                                            \
      _Pragma("unroll") for (int c = 0; c < CH_W; ++c) {                                                      \
        cR[m][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR[m], bR[c], cR[m][c], 0, 0, 0);                     \
        c3[m][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI[m], bI[c], c3[m][c], 0, 0, 0);                     \
        cI[m][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[m], dB[c], cI[m][c], 0, 0, 0);                     \
      }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
#pragma clang loop unroll(disable)
  for (int it = 0; it < nit; it += 2) {
    CH_BU_STEP(xaR, xaI, xbR, xbI, yaR, yaI, ybR, ybI, it + 1)
    const int nx2 = (it + 2 < nit) ? (it + 2) : (it + 1);
    CH_BU_STEP(yaR, yaI, ybR, ybI, xaR, xaI, xbR, xbI, nx2)
  }
#undef CH_BU_STEP
  // T[P][Q0+c] -= sum  (Re -= sum a conj(b) real part;  the second accumulator holds minus the imaginary part); the
  // eight values of a tile are requested together
#pragma unroll
  for (int m = 0; m < RB; ++m) {
    const int P = Pg + RS * m;
#pragma unroll
    for (int c = 0; c < CH_W; ++c) {
      if (c >= Wc || Q0 + c > P) continue;   // (tile above the diagonal / beyond the last column)
      double* Ct = T + tile_off(P, Q0 + c) + lq * 16 + lcol;
      double oR[4], oI[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { oR[r] = Ct[64 * r]; oI[r] = Ct[256 + 64 * r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Ct[64 * r] = oR[r] - (cR[m][c][r] + c3[m][c][r]);
        Ct[256 + 64 * r] = oI[r] + (cI[m][c][r] + (cR[m][c][r] - c3[m][c][r]));
      }
    }
  }
}

// T: tiles; WT: nt * 2*TL_TILE doubles (W^H of every diagonal tile); D0: 16*nt doubles (original diagonal, for the
// collapsed-pivot check); lds: CH_LDS_DOUBLES; xs: 4*Kc doubles out (pair_n: column order of the caller, see the back
// substitution)
template <int NW>
__device__ inline void tile_cholesky_memory(double* __restrict__ T, double* __restrict__ WT, double* __restrict__ D0,
                                            int nt, int Kc, int nbk, double* lds, double* xs, int* fault,
                                            unsigned long long* dbg = nullptr, int pair_n = -1) {
  const int tid = threadIdx.x, lane = tid & 63, lcol = lane & 15, lq = lane >> 4;
  // phase stamps of thread 0, slots 4-9 (diagnostic build -DEAQHM_EXPERIMENT_STAMPS only — compiled in they cost the
  // large-frame kernel 5 % through its register allocation; tools/phase_probe_big.py)
#ifndef EAQHM_EXPERIMENT_STAMPS
#define CH_STAMP(ph) do { } while (0)
#else
  unsigned long long t_prev = (dbg && tid == 0) ? __builtin_amdgcn_s_memtime() : 0ull;
#define CH_STAMP(ph)                                              \
  do {                                                            \
    if (dbg && tid == 0) {                                        \
      const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
      atomicAdd(dbg + (ph), t_now - t_prev);                      \
      t_prev = t_now;                                             \
    }                                                             \
  } while (0)
#endif
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RB = (NW == 8) ? CH_RB : 1;   // tile rows per wave and register group of the block update (NW * RB rows per group)
  constexpr int MB = (8 * CH_MB) / NW;        // tiles per wave, column and register group (48 rows per group)
  static_assert(NW == 8 || NW == 12 || NW == 16, "tile_cholesky_memory: 8, 12 or 16 waves");
  double* Dc = lds;
  double* Zc = Dc + DG_TILE;
  double* WtR = Zc + DG_TILE;
  double* WtI = WtR + TL_TILE;
  double* LdR = WtI + TL_TILE;
  double* LdI = LdR + TL_TILE;
  double* trb = LdI + TL_TILE + (size_t)wave * 2 * TL_TILE;   // this wave's transposition buffer
  double* zv = LdI + TL_TILE + NW * 2 * TL_TILE;
  double* xv = zv + 2 * 16 * CH_NTMAX;
  // original diagonal of the whole system (the collapsed-pivot check of diag_coop compares against it)
  for (int q = tid; q < 16 * nt; q += blockDim.x) {
    const int Qd = q >> 4, i = q & 15;
    D0[q] = T[tile_off(Qd, Qd) + i * 16 + i];
  }
  __syncthreads();

  for (int Q0 = 0; Q0 < nt; Q0 += CH_W) {
   const int Wc = (nt - Q0 < CH_W) ? (nt - Q0) : CH_W;
#ifdef EAQHM_EXPERIMENT_NOBLOCKUPD
   if (false) {
#else
   if (Q0 > 0) {
#endif
    // ---- (1) block update of columns Q0 .. Q0+Wc-1 from the columns before the block.  Rows P = Q0 + wave + 8 x,
    //      CH_RB of them per register group; three accumulators per tile, three real products per complex one.
    const int ngr = (nt - Q0 + NW * RB - 1) / (NW * RB);
    for (int g = 0; g < ngr; ++g) {
      const int Pg = Q0 + wave + NW * RB * g;
      if (Pg >= nt) continue;
      if (RB == 2 && Pg + NW < nt) chol_block_unit<2, NW>(T, Q0, Wc, Pg, lq, lcol);
      else chol_block_unit<1, NW>(T, Q0, Wc, Pg, lq, lcol);
    }
    __syncthreads();   // the block's tiles are up to date with every column before the block
    CH_STAMP(4);
   }
   for (int Q = Q0; Q < Q0 + Wc; ++Q) {
    // ---- (2) column Q: tiles P = Q + wave + NW m, in groups of MB per wave (registers); the first group holds the
    // diagonal tile, which is factorised before any panel tile is finished.  Only the block's own earlier columns
    // are still to be subtracted.
    const double* dref = D0 + 16 * Q;
    const int ngroups = (nt - Q + NW * MB - 1) / (NW * MB);
    for (int grp = 0; grp < ngroups; ++grp) {
      const int Pb = Q + wave + NW * MB * grp;   // this wave's first tile of the group
      d4 p1[MB], p2[MB], p3[MB];
#pragma unroll
      for (int m = 0; m < MB; ++m) { p1[m] = (d4){0, 0, 0, 0}; p2[m] = (d4){0, 0, 0, 0}; p3[m] = (d4){0, 0, 0, 0}; }
      for (int j = Q0; j < Q; ++j) {
        const double* Bt = T + tile_off(Q, j);
        double bR[4], bI[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const int o = (4 * ks + lq) * 16 + lcol; bR[ks] = Bt[o]; bI[ks] = Bt[256 + o]; }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const int P = Pb + NW * m;
          if (P >= nt) continue;
          const double* At = T + tile_off(P, j);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const int o = (4 * ks + lq) * 16 + lcol;
            const double aR = At[o], aI = At[256 + o];
            p1[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR[ks], p1[m], 0, 0, 0);
            p2[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI[ks], p2[m], 0, 0, 0);
            p3[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(aR + aI, bI[ks] - bR[ks], p3[m], 0, 0, 0);
          }
        }
      }
      CH_STAMP(14);
      // C = T[P][Q] - sum:  Re -= P1 + P2,  Im += P3 + P1 - P2   (kept in p1 / p3).  The eight values of the NEXT tile are
      // requested before this one is combined: one exposed round trip per column instead of one per tile: 515 -> 500 ms.
      // (The same arrangement for the A tiles of the j-loop above measured slower, 504.7 vs 499.7 ms, and for the
      // write-back of the block update no different, 500.1.)
      {
        double oR[4], oI[4];
        {
          const double* Ct = T + tile_off((Pb < nt) ? Pb : (nt - 1), Q) + lq * 16 + lcol;
#pragma unroll
          for (int r = 0; r < 4; ++r) { oR[r] = Ct[64 * r]; oI[r] = Ct[256 + 64 * r]; }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const int P = Pb + NW * m, Pn = P + NW;
          double nR[4] = {0, 0, 0, 0}, nI[4] = {0, 0, 0, 0};
          if (m + 1 < MB && Pn < nt) {
            const double* Ct = T + tile_off(Pn, Q) + lq * 16 + lcol;
#pragma unroll
            for (int r = 0; r < 4; ++r) { nR[r] = Ct[64 * r]; nI[r] = Ct[256 + 64 * r]; }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (P < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double cr = oR[r] - (p1[m][r] + p2[m][r]);
              const double ci = oI[r] + (p3[m][r] + (p1[m][r] - p2[m][r]));
              p1[m][r] = cr; p3[m][r] = ci;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) { oR[r] = nR[r]; oI[r] = nI[r]; }
        }
      }
      CH_STAMP(5);
      if (grp == 0) {
        if (wave == 0) {   // the diagonal tile is this wave's first tile of the first group
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            Dc[2 * ((lq + 4 * r) * DG_LD + lcol)] = p1[0][r];
            Dc[2 * ((lq + 4 * r) * DG_LD + lcol) + 1] = p3[0][r];
          }
        }
        diag_init(Zc, tid);
        __syncthreads();
        CH_STAMP(15);
        // the last tile row is the right-hand side (row 0) plus identity padding: no real unknown there
#ifndef EAQHM_EXPERIMENT_NOCHOLDIAG
        diag_coop(Dc, Zc, WtR, WtI, LdR, LdI, tid, dref, (Q == nt - 1) ? 0 : 16, fault);   // ends with a barrier
#endif
        for (int q = tid; q < 2 * TL_TILE; q += blockDim.x) WT[(size_t)Q * 2 * TL_TILE + q] = WtR[q];   // WtR | WtI contiguous
        CH_STAMP(6);
      }
      // panel tiles: X = C W^H (three real products), stored k-major
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int P = Pb + NW * m;
        if (P >= nt || P == Q) continue;
        double* tr = trb;
        double* ti = trb + TL_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // C[row = lq+4r][col = lcol] -> tmp[k = col][i = row]
          tr[lcol * TL_LD + lq + 4 * r] = p1[m][r];
          ti[lcol * TL_LD + lq + 4 * r] = p3[m][r];
        }
        __builtin_amdgcn_wave_barrier();
        d4 x1 = (d4){0, 0, 0, 0}, x2 = (d4){0, 0, 0, 0}, x3 = (d4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int o = (4 * ks + lq) * TL_LD + lcol;
          const double aR = tr[o], aI = ti[o], bR = WtR[o], bI = WtI[o];
          x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aR, bR, x1, 0, 0, 0);
          x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(aI, bI, x2, 0, 0, 0);
          x3 = __builtin_amdgcn_mfma_f64_16x16x4f64(aR + aI, bR + bI, x3, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        double* Lt = T + tile_off(P, Q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // L[i = lq+4r][k = lcol] -> [k][i]
          Lt[lcol * 16 + lq + 4 * r] = x1[r] - x2[r];
          Lt[256 + lcol * 16 + lq + 4 * r] = x3[r] - (x1[r] + x2[r]);
        }
      }
      CH_STAMP(7);
    }
    __syncthreads();   // factor column visible to every wave; Dc / Wt reusable
    CH_STAMP(8);
   }
  }

  // ---- back substitution  L^H x = y,  y = conj(row 0 of the last tile row)
  for (int q = tid; q < 16 * (nt - 1); q += blockDim.x) {
    const int Qt = q >> 4, c = q & 15;
    const double* Lt = T + tile_off(nt - 1, Qt);
    zv[2 * q] = Lt[c * 16];
    zv[2 * q + 1] = -Lt[256 + c * 16];
  }
  for (int q = tid; q < 4 * Kc; q += blockDim.x) xs[q] = 0.0;
  __syncthreads();
  for (int P = nt - 2; P >= 0; --P) {
    if (tid < 256) {  // x_P = W_PP^H z_P : thread (i, k) takes one term of row i, 16-lane shuffle reduction
      const int i = tid >> 4, k = tid & 15;
      double xr = 0, xi = 0;
      if (k >= i) {   // W^H is upper triangular
        const double wr = WT[(size_t)P * 2 * TL_TILE + i * TL_LD + k], wi = WT[(size_t)P * 2 * TL_TILE + TL_TILE + i * TL_LD + k];
        const double zr = zv[2 * (16 * P + k)], zi = zv[2 * (16 * P + k) + 1];
        xr = wr * zr - wi * zi;
        xi = wr * zi + wi * zr;
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { xr += __shfl_xor(xr, o); xi += __shfl_xor(xi, o); }
      if (k == 0) {
        xv[2 * i] = xr; xv[2 * i + 1] = xi;
        const int q = (P < nbk) ? (16 * P + i) : (16 * (P - nbk) + i);   // amplitudes, then slopes
        if (q < Kc) {
          // pair_n = n >= 0: the caller's columns are ordered [neg 0, pos 0, neg 1, pos 1, ..., DC]; xs is [neg | DC | pos]
          const int qo = (pair_n < 0) ? q : ((q == 2 * pair_n) ? pair_n : ((q & 1) ? (pair_n + 1 + (q >> 1)) : (q >> 1)));
          const int d = (P < nbk) ? qo : (Kc + qo);
          xs[2 * d] = xr; xs[2 * d + 1] = xi;
        }
      }
    }
    __syncthreads();
    for (int q = tid; q < 16 * P; q += blockDim.x) {   // z_Q -= L[P][Q]^H x_P
      const int Qt = q >> 4, c = q & 15;
      const double* Lt = T + tile_off(P, Qt) + c * 16;
      double sr = 0, si = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double lr = Lt[i], li = Lt[256 + i], xr = xv[2 * i], xi = xv[2 * i + 1];
        sr += lr * xr + li * xi;
        si += lr * xi - li * xr;
      }
      zv[2 * q] -= sr;
      zv[2 * q + 1] -= si;
    }
    __syncthreads();
  }
  CH_STAMP(9);
#undef CH_STAMP
}

}  // namespace eaqhm
