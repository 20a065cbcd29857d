"""ctypes binding of libeaqhm_hip.so (C ABI: include/eaqhm_hip.h).

There is NO CPU fallback: if the library is missing or no MI355X is visible, every entry point of
the package raises `HipUnavailable`.  Device buffers are torch-ROCm tensors; only their raw
`data_ptr()` crosses the ABI.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EAQHM_LIB: A/B measurements with an alternative build of the same library (tools/); the product default is in-tree
LIB_PATH = os.environ.get("EAQHM_LIB") or os.path.join(_HERE, "csrc", "libeaqhm_hip.so")

# every symbol include/eaqhm_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I32, _I64, _F64 = C.c_int32, C.c_int64, C.c_double
SYMBOLS = (
    ("eaqhm_ctx_create", C.c_int, [C.POINTER(_P), C.c_int]),
    ("eaqhm_ctx_destroy", C.c_int, [_P]),
    ("eaqhm_set_stream", C.c_int, [_P, _P]),
    ("eaqhm_sync", C.c_int, [_P]),
    ("eaqhm_last_error", C.c_char_p, [_P]),
    ("eaqhm_set_option", C.c_int, [_P, _I32, _I32]),
    ("eaqhm_debug_read", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("eaqhm_device_info", C.c_int, [_P, C.POINTER(_I32)]),
    ("eaqhm_ls_faults", C.c_int, [_P, C.POINTER(_I32)]),
    ("eaqhm_frame_prep", C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _P, _I32, _P, _P, _P, _P]),
    ("eaqhm_ls_batch", C.c_int, [_P, _I32, _P, _I64, _F64, _P, _P, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                  _I32, _I32, _I32, _F64, _F64, _P, _P, _P]),
    ("eaqhm_ls_explicit", C.c_int, [_P, _P, _I32, _P, _P, _P, _I32, _P, _F64, _P, _P]),
    ("eaqhm_phase_integrate", C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P]),
    ("eaqhm_spline_solve", C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P]),
    ("eaqhm_spline_solve_range", C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    ("eaqhm_eval_synth", C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _F64, _I64, _I64, _I64, _I64, _I64, _P, _F64,
                                    _P, _P, _I64, _I64, _P, _P, _P, _P]),
    ("eaqhm_eval_partials_len", _I64, [_I64, _I64, _I32]),
)


class HipUnavailable(RuntimeError):
    """The HIP library or the GPU is missing.  The package has no CPU path."""


_lib = None


def load_library():
    """Load libeaqhm_hip.so and bind every symbol of the header (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipUnavailable(
            "libeaqhm_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipUnavailable("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, res, args in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipUnavailable("libeaqhm_hip.so lacks symbol %s (stale build?)" % name) from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("device buffers must be contiguous GPU tensors")
    return t.data_ptr()


class Context:
    """One eaqhm_ctx bound to a device and to torch's current stream on it."""

    def __init__(self, device_index=0):
        import torch
        if not torch.cuda.is_available():
            raise HipUnavailable("no ROCm GPU visible (torch.cuda.is_available() is False); "
                                 "this package has no CPU fallback")
        self.lib = load_library()
        self.torch = torch
        self.device = torch.device("cuda", device_index)
        h = _P()
        rc = self.lib.eaqhm_ctx_create(C.byref(h), device_index)
        if rc != 0:
            raise HipUnavailable("eaqhm_ctx_create failed with code %d" % rc)
        self.h = h
        info = (_I32 * 4)()
        self._ck(self.lib.eaqhm_device_info(self.h, info))
        self.n_cu, self.lds_bytes, self.clock_khz, self.abi_version = [int(v) for v in info]
        v = os.environ.get("EAQHM_LS_VARIANT")          # A/B knob for measurements (include/eaqhm_hip.h)
        if v:
            self.set_option(1, int(v))
        self.bind_stream()

    def bind_stream(self):
        s = self.torch.cuda.current_stream(self.device)
        self._ck(self.lib.eaqhm_set_stream(self.h, _P(s.cuda_stream)))

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.eaqhm_last_error(self.h)
            raise RuntimeError("libeaqhm_hip error %d: %s" % (rc, msg.decode() if msg else "?"))

    def close(self):
        if getattr(self, "h", None):
            self.lib.eaqhm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._ck(self.lib.eaqhm_set_option(self.h, key, value))

    def debug_read(self):
        out = (C.c_uint64 * 16)()
        self._ck(self.lib.eaqhm_debug_read(self.h, out))
        return [int(v) for v in out]

    def ls_faults(self):
        """(LS systems whose Cholesky broke down, stalled diagonal pipelines, frames dropped because their window was not
        resident) since the last read (waits for the stream, clears the counts)."""
        n = (_I32 * 3)()
        self._ck(self.lib.eaqhm_ls_faults(self.h, n))
        return int(n[0]), int(n[1]), int(n[2])

    def sync(self):
        self._ck(self.lib.eaqhm_sync(self.h))

    # ---- thin wrappers (argument order = header order)
    def frame_prep(self, fm_cur, L, track_t0, track_len, Kmax, frame_c, n_frames, ncol, cols, seeded, any_seed):
        self._ck(self.lib.eaqhm_frame_prep(self.h, _ptr(fm_cur), L, track_t0, track_len, Kmax, _ptr(frame_c), n_frames,
                                           _ptr(ncol), _ptr(cols), _ptr(seeded), _ptr(any_seed)))

    def ls_batch(self, mode, s, L, fs, am_cur, fm_cur, track_t0, track_len, Kmax, frame_inst, frame_c, frame_wl,
                 frame_f0, frame_K, ncol, cols, seeded, any_seed, n_frames, wl_max, a_iter, f0_stale, f0min, records,
                 raw_amp=None, raw_slope=None):
        self._ck(self.lib.eaqhm_ls_batch(self.h, mode, _ptr(s), L, float(fs), _ptr(am_cur), _ptr(fm_cur), track_t0,
                                         track_len, Kmax,
                                         _ptr(frame_inst), _ptr(frame_c), _ptr(frame_wl), _ptr(frame_f0),
                                         _ptr(frame_K), _ptr(ncol), _ptr(cols), _ptr(seeded), _ptr(any_seed),
                                         n_frames, wl_max, a_iter, float(f0_stale), float(f0min),
                                         _ptr(records), _ptr(raw_amp), _ptr(raw_slope)))

    def ls_explicit(self, s, N, am, fm, f0range, Kc, window, fs, out_amp, out_slope):
        self._ck(self.lib.eaqhm_ls_explicit(self.h, _ptr(s), N, _ptr(am), _ptr(fm), _ptr(f0range), Kc, _ptr(window),
                                            float(fs), _ptr(out_amp), _ptr(out_slope)))

    def phase_integrate(self, omega, ph, knots, n_knots, first, last, out):
        self._ck(self.lib.eaqhm_phase_integrate(self.h, _ptr(omega), _ptr(ph), _ptr(knots), n_knots, first, last, _ptr(out)))

    def spline_solve(self, records, No_ti, Kmax, step, code, mom, i_lo=0, i_hi=None):
        i_hi = No_ti if i_hi is None else i_hi
        self._ck(self.lib.eaqhm_spline_solve_range(self.h, _ptr(records), No_ti, Kmax, step, i_lo, i_hi, _ptr(code),
                                                   _ptr(mom)))

    def eval_synth(self, records, code, mom, No_ti, Kmax, step, fs, L, t_lo, t_hi, s_lo, s_hi, target, std_det,
                   am_out, fm_out, track_t0, track_len, ph_knot, s_hat, partials, sums_out):
        self._ck(self.lib.eaqhm_eval_synth(self.h, _ptr(records), _ptr(code), _ptr(mom), No_ti, Kmax,
                                           step, float(fs), L, t_lo, t_hi, s_lo, s_hi, _ptr(target), float(std_det),
                                           _ptr(am_out), _ptr(fm_out), track_t0, track_len, _ptr(ph_knot), _ptr(s_hat),
                                           _ptr(partials), _ptr(sums_out)))

    def eval_partials_len(self, t_lo, t_hi, step):
        return int(self.lib.eaqhm_eval_partials_len(t_lo, t_hi, step))
