/* eaqhm_hip.h — C ABI of libeaqhm_hip.so: the MI355X (gfx950) implementation of the eaQHM per-frame
 * analysis hot path.
 *
 * The reference (Antibas/eaQHM-analysis-and-synthesis-in-Python) has no FFI: its boundary for this path
 * is the Python function eaQHMAnalysisAndSynthesis (functions.py:35-418) and the inner seams
 * iqhmLS_complexamps (functions.py:420-470), eaqhmLS_complexamps (functions.py:472-535) and
 * phase_integr_interpolation (functions.py:537-575).  Each entry point below names the reference lines
 * it replaces.  The host side (Python, ctypes) lives in eaqhm-analysis-and-synthesis-in-python_amd/.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless its name starts with h_ (host); buffers are owned by the
 *    caller (the Python host allocates them as torch-ROCm tensors and passes tensor.data_ptr());
 *  - all work is enqueued on the stream given to eaqhm_set_stream (a hipStream_t passed as void*;
 *    NULL = the default stream) and is asynchronous; eaqhm_sync waits for it;
 *  - every function returns 0 on success or a negative EAQHM_E* code; eaqhm_last_error gives the text;
 *    nothing throws, nothing frees caller memory;
 *  - floating point is IEEE double everywhere (the reference is float64/complex128 throughout).
 *
 * Layouts ("harmonic-major" = the reference's (L, Kmax) arrays transposed so time is contiguous)
 *  - s, target, s_hat       double[L]
 *  - am_cur, fm_cur         double[Kmax][track_len] dense tracks of the previous adaptation (functions.py:159-160,
 *                                                   :337-338, :375, :383) for the samples [track_t0, track_t0 +
 *                                                   track_len) of the file: the whole file (0, L), one rank's time
 *                                                   range plus halo, or one time block of a long file (the reference
 *                                                   keeps seven such (L, Kmax) arrays resident).  A frame window
 *                                                   [c-wl-1, c+wl] handed to eaqhm_ls_batch must lie inside it.
 *  - records                double[No_ti][3*Kmax+1] one row per analysis instant: |a_k| (Kmax), f_k (Kmax),
 *                                                   arg a_k (Kmax) written at functions.py:316-324, then the
 *                                                   DC term a0 (functions.py:303).  Rows of one contiguous
 *                                                   range of instants are one contiguous block, which is what
 *                                                   the per-adaptation all-gather across GPUs moves.
 *  - frame tables           one entry per ANALYSED frame (functions.py:180-181 true), ascending in time
 */
#ifndef EAQHM_HIP_H
#define EAQHM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAQHM_OK 0
#define EAQHM_EINVAL (-1)  /* bad argument (shape / range)        */
#define EAQHM_EHIP (-2)    /* a HIP runtime call failed            */
#define EAQHM_ENOMEM (-3)  /* scratch allocation failed            */

typedef struct eaqhm_ctx eaqhm_ctx;

/* life cycle ------------------------------------------------------------------------------------ */
int eaqhm_ctx_create(eaqhm_ctx** out, int device);
int eaqhm_ctx_destroy(eaqhm_ctx* ctx);
int eaqhm_set_stream(eaqhm_ctx* ctx, void* hip_stream);
int eaqhm_sync(eaqhm_ctx* ctx);
const char* eaqhm_last_error(eaqhm_ctx* ctx);
/* tuning knobs (for A/B measurements; defaults are the fastest validated choice)
 *   EAQHM_OPT_LS_VARIANT: 2 = MFMA Gramian + tile Cholesky through memory for every frame (the large-frame kernel),
 *                         3 = Gramian and tile Cholesky on chip for frames of <= 13 tile rows (Kc <= 103), the
 *                             large-frame kernel for the rest (default) */
#define EAQHM_OPT_LS_VARIANT 1
#define EAQHM_OPT_DEBUG_KEEP 2   /* 1: accumulate the in-kernel phase stamps across launches; 2: also time diag_D */
int eaqhm_set_option(eaqhm_ctx* ctx, int32_t key, int32_t value);
/* diagnostics: shader-clock cycles per phase of the LS tile kernel summed over frames (thread 0 of each
 * workgroup): {setup, basis build, contraction, factorisation total..., see csrc/eaqhm_ls_tile.hip STAMP} */
int eaqhm_debug_read(eaqhm_ctx* ctx, uint64_t h_out[16]);
/* singular systems: the reference aborts with numpy.linalg.LinAlgError from inv() when a frame's normal matrix is
 * singular — an exactly zero LU pivot (functions.py:465, :530); an ill-conditioned system is solved and returned.  The
 * kernels factorise by Cholesky; what corresponds to the exact zero is a BREAKDOWN of the factorisation: a pivot that
 * is not positive or has fallen to <= 2.5e-13 (order x eps) of its original diagonal entry, as two identical basis
 * columns give.  Such frames are counted in a device counter (h_count[0]); ill-conditioned but factorisable systems are
 * solved like the reference solves them.  h_count[1] counts diagonal-tile pipelines whose internal hand-shake timed
 * out — never nonzero unless the library has a bug; the host raises RuntimeError for it, LinAlgError for h_count[0].
 * h_count[2] counts frames whose analysis window [c-wl-1, c+wl] was not inside the signal / the resident track window
 * handed to eaqhm_ls_batch: such a frame is dropped (its record row is not written) instead of being read out of bounds;
 * the host raises ValueError.  eaqhm_ls_faults waits for the stream, returns the counts since the last read and clears
 * them; eaqhm_eval_synth also reports (and clears) them in sums_out[4..6], so the adaptation loop needs no extra
 * device->host read. */
int eaqhm_ls_faults(eaqhm_ctx* ctx, int32_t h_count[3]);
/* library / device facts: fills {n_cu, lds_bytes, clock_khz, abi_version} */
int eaqhm_device_info(eaqhm_ctx* ctx, int32_t h_info[4]);

/* adaptation >= 1 frame set-up ---------------------------------------------------------------------
 * Replaces functions.py:202-213: the active-slot list of each frame (indices of nonzero
 * fm_current[c,:]) and the empty-row seeding flag (slot 0 <- 140 Hz / 10e-4).  The seeding WRITE of the
 * reference is not performed: `seeded[c]` (uint8[L], zeroed here) marks the rows and the LS kernel
 * applies "visible to frames at or after c" itself, which reproduces the sequential loop exactly.
 *   ncol[f]            number of active slots of frame f
 *   cols[f*Kmax + j]   j-th active slot (ascending)
 *   any_seed           int32[1], nonzero if any frame was seeded                                     */
int eaqhm_frame_prep(eaqhm_ctx* ctx, const double* fm_cur, int64_t L, int64_t track_t0, int64_t track_len, int32_t Kmax,
                     const int32_t* frame_c, int32_t n_frames, int32_t* ncol, int32_t* cols, uint8_t* seeded,
                     int32_t* any_seed);

/* the per-frame least squares, batched over frames ---------------------------------------------------
 * Replaces, for every analysed frame of one adaptation:
 *   adaptation 0 (mode 0): functions.py:187-197 + iqhmLS_complexamps (functions.py:420-470)
 *   adaptation>=1 (mode 1): functions.py:244-295 (track windows, zero-gap fill, negative/DC/positive
 *                           column layout) + eaqhmLS_complexamps (functions.py:472-535)
 *   both: frequency correction (functions.py:297), slot scatter (:299-301), acceptance test and
 *         frame-centre writes (:303-324).
 * Inputs
 *   frame_inst[f] instant index i (row of the records), frame_c[f] 0-based centre sample (tith-1),
 *   frame_wl[f] half window length (functions.py:191), frame_f0[f] / frame_K[f] adaptation-0 pitch and
 *   harmonic count (functions.py:185-187; ignored in mode 1), ncol/cols/seeded/any_seed from
 *   eaqhm_frame_prep (mode 1; may be NULL in mode 0), wl_max >= max(frame_wl) (sizes the per-workgroup
 *   scratch), a_iter the adaptation number (h = f0/(a+1),
 *   functions.py:310), f0_stale the pitch of the last frame of adaptation 0 (used for every frame when
 *   a_iter >= 1, functions.py:310/:321 quirk), f0min (functions.py:321).
 * Outputs
 *   records[i][...] for the frames' instants (the whole row of 3*Kmax+1 values is written),
 *   optional raw_amp / raw_slope: double[n_frames][2*(2*Kmax+1)] interleaved complex LS solutions in
 *   column order [negative block | DC | positive block] (NULL to skip) — what the two seam functions
 *   return.                                                                                           */
int eaqhm_ls_batch(eaqhm_ctx* ctx, int32_t mode, const double* s, int64_t L, double fs, const double* am_cur,
                   const double* fm_cur, int64_t track_t0, int64_t track_len, int32_t Kmax,
                   const int32_t* frame_inst, const int32_t* frame_c,
                   const int32_t* frame_wl, const double* frame_f0, const int32_t* frame_K, const int32_t* ncol,
                   const int32_t* cols, const uint8_t* seeded, const int32_t* any_seed, int32_t n_frames,
                   int32_t wl_max, int32_t a_iter, double f0_stale, double f0min, double* records, double* raw_amp,
                   double* raw_slope);

/* the two LS seams with explicit matrices, one frame ------------------------------------------------
 * eaqhm_ls_explicit: eaqhmLS_complexamps(s, am, fm, window, fs) (functions.py:472-535) when `fm` is
 * non-NULL — am, fm are double[N][Kc] row-major; iqhmLS_complexamps(s, f0range, window, fs)
 * (functions.py:420-470) when `fm` is NULL and `f0range` (double[Kc]) is given.
 * out_amp / out_slope: double[2*Kc] interleaved complex.                                            */
int eaqhm_ls_explicit(eaqhm_ctx* ctx, const double* s, int32_t N, const double* am, const double* fm,
                      const double* f0range, int32_t Kc, const double* window, double fs, double* out_amp,
                      double* out_slope);

/* the third inner seam, stand-alone ---------------------------------------------------------------
 * phase_integr_interpolation(fm_recon, ph_recon, indices) (functions.py:537-575): `omega` = 2*pi/fs * fm_recon
 * and `ph` are dense columns, `knots` (int32[n_knots], ascending, any spacing) the knot samples with
 * first = knots[0], last = knots[n_knots-1]; out = double[last-first+1], the dense phase on [first, last]. */
int eaqhm_phase_integrate(eaqhm_ctx* ctx, const double* omega, const double* ph, const int32_t* knots,
                          int32_t n_knots, int32_t first, int32_t last, double* out);

/* interpolation stage 1: segments + spline systems ---------------------------------------------------
 * Replaces the knot bookkeeping and the not-a-knot cubic solves of functions.py:340 (a0, all instants)
 * and :346-371 (per harmonic: runs of consecutive accepted instants, cubic through the knots).
 *   code[i][k]  uint8: 0 not accepted, 1 isolated accepted instant, 2 member of a run of >= 4 knots,
 *               16 + 4*m + pos for runs of m = 2 or 3 knots (pos = position inside the run)
 *   mom[i][k]   double[No_ti][Kmax+1] second derivatives of the fm splines (column Kmax: the a0 spline) */
int eaqhm_spline_solve(eaqhm_ctx* ctx, const double* records, int32_t No_ti, int32_t Kmax, int32_t step,
                       uint8_t* code, double* mom);
/* the same for the instants [i_lo, i_hi) only (a rank that evaluates only its own time range: the moments are
 * local sums, so the rest of `code` / `mom` is neither read nor written, except the run codes of instants 0..3) */
int eaqhm_spline_solve_range(eaqhm_ctx* ctx, const double* records, int32_t No_ti, int32_t Kmax, int32_t step,
                             int32_t i_lo, int32_t i_hi, uint8_t* code, double* mom);

/* interpolation stage 2 + synthesis + SRER -----------------------------------------------------------
 * Replaces functions.py:364 (linear am), :367-371 (cubic fm, incl. the <4-knot padded case),
 * :373 + phase_integr_interpolation (functions.py:537-575), :375 (next-iteration frequency from the
 * unwrapped phase), :383 (am_current), :385 (additive synthesis) and :388 (SRER).
 * Samples [t_lo, t_hi) are produced (the whole signal when 0, L); the error sums cover [s_lo, s_hi)
 * inside that range (a rank of a time-sharded run produces its range plus a halo but sums only its own).
 *   am_out, fm_out   double[Kmax][track_len]: next adaptation's am_current / fm_current for the samples
 *                    [track_t0, track_t0 + track_len) >= [t_lo, t_hi); both NULL: no track output (a long file's
 *                    synthesis pass: its tracks are regenerated block by block from the records)
 *   ph_knot          double[No_ti][Kmax] dense phase at the instants (what functions.py:411 packs)
 *   s_hat            double[L]; NULL: tracks only — no synthesis, no ph_knot, no error sums (target, ph_knot,
 *                    partials, sums_out may then be NULL too)
 *   partials         8-byte words, eaqhm_eval_partials_len of them: per-block error sums
 *   sums_out         double[16]: {sum d, sum d^2, n, SRER dB, LS breakdowns, stalled pipelines and dropped frames since the
 *                    last read (see eaqhm_ls_faults), -, then eight int64 bit patterns} with d = target - s_hat over
 *                    [s_lo,s_hi).  The int64 words are the same sums in fixed point — three base-2^32 limbs of
 *                    d*2^60, three of d^2*2^64, the number of samples with |d| >= 2^20 or non-finite, 0 — which add
 *                    up exactly over blocks, ranks and time blocks, so the SRER (functions.py:388, the input of the
 *                    stop rule :394) does not depend on how the file was split; sums_out[3] is that SRER for this
 *                    call's range alone (std_det: functions.py:161).                                  */
int eaqhm_eval_synth(eaqhm_ctx* ctx, const double* records, const uint8_t* code, const double* mom,
                     int32_t No_ti, int32_t Kmax, int32_t step, double fs, int64_t L, int64_t t_lo, int64_t t_hi,
                     int64_t s_lo, int64_t s_hi, const double* target, double std_det, double* am_out,
                     double* fm_out, int64_t track_t0, int64_t track_len, double* ph_knot, double* s_hat,
                     double* partials, double* sums_out);
/* number of 8-byte words `partials` must hold for a given range */
int64_t eaqhm_eval_partials_len(int64_t t_lo, int64_t t_hi, int32_t step);

#ifdef __cplusplus
}
#endif
#endif
