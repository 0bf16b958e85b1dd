"""Python host mirror of the reference's dense iterative solver over the C ABI.

Same factory and method names as matrix_webgl.makeSORIterative(spec)
(matrix_webgl.js:35-711): vec_length, vec_height, set_matrix, set_b, init_vector, mv_product,
solve, x_result_tex.  Binds include/fusionsor.h (libfusionpic.so) with ctypes; there is no CPU
path.  Used by tests/, bench.py; the shipped host language is JavaScript
(fusion-sim_amd/js/matrix_native.js over the N-API addon).
"""
import ctypes

import numpy as np

from . import F32, F64, FusionPicError, load_library

X_RESULT, X_GUESS, X_STATS, VEC_C, VEC_B = range(5)

ABI_FUNCTIONS = [
    "fsor_last_error", "fsor_abi_version", "fsor_create", "fsor_destroy", "fsor_dims", "fsor_set_matrix", "fsor_set_b",
    "fsor_init_vector", "fsor_prepare", "fsor_iterate", "fsor_solve", "fsor_read_vector", "fsor_read_iteration_matrix",
    "fsor_device_buffer", "fsor_set_stream", "fsor_sync", "fsor_profile", "fsor_get_stats", "fsor_reset_stats",
]


class SorSpec(ctypes.Structure):
    _fields_ = [("n_power", ctypes.c_int32), ("device", ctypes.c_int32), ("relaxation", ctypes.c_double),
                ("natural_rows", ctypes.c_int32), ("reserved", ctypes.c_int32 * 5)]


class SorResult(ctypes.Structure):
    _fields_ = [("correlation", ctypes.c_double), ("diff", ctypes.c_double), ("iterations", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class SorStats(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_uint64), ("seconds_iterate", ctypes.c_double), ("matrix_bytes", ctypes.c_uint64)]


_bound = False


def _lib():
    global _bound
    lib = load_library()
    if not _bound:
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.fsor_last_error.restype = ctypes.c_char_p
        lib.fsor_last_error.argtypes = [vp]
        lib.fsor_create.argtypes = [ctypes.POINTER(SorSpec), ctypes.POINTER(vp)]
        lib.fsor_destroy.argtypes = [vp]
        lib.fsor_dims.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)]
        for f in ("fsor_set_matrix", "fsor_set_b", "fsor_init_vector"):
            getattr(lib, f).argtypes = [vp, vp, ci]
        for f in ("fsor_prepare", "fsor_sync", "fsor_reset_stats"):
            getattr(lib, f).argtypes = [vp]
        lib.fsor_iterate.argtypes = [vp, ctypes.c_int32]
        lib.fsor_profile.argtypes = [vp, ci]
        lib.fsor_solve.argtypes = [vp, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.POINTER(SorResult), vp]
        lib.fsor_read_vector.argtypes = [vp, ci, vp]
        lib.fsor_read_iteration_matrix.argtypes = [vp, vp]
        lib.fsor_device_buffer.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        lib.fsor_set_stream.argtypes = [vp, vp]
        lib.fsor_get_stats.argtypes = [vp, ctypes.POINTER(SorStats)]
        _bound = True
    return lib


def _validate(obj, props):
    """util.validate_object with 'number' / [,'number'] controls (utilities.js:11-127)."""
    for key, optional in props:
        if key not in obj or obj[key] is None:
            if optional:
                continue
            raise FusionPicError(-1, "." + key + " <- Non-optional property is undefined!")
        if isinstance(obj[key], bool) or not isinstance(obj[key], (int, float, np.integer, np.floating)):
            raise FusionPicError(-1, "." + key + " <- Property does not match any given possible types!")


def _host(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a, F32
    return np.ascontiguousarray(a, dtype=np.float64), F64


class SORIterative:
    """The object makeSORIterative(spec) returns (matrix_webgl.js:42-709)."""

    def __init__(self, spec, device=0, compat=True):
        _validate(spec, (("n_power", False), ("relaxation", True)))
        self._lib = _lib()
        cs = SorSpec(n_power=int(spec["n_power"]), device=int(device), relaxation=float(spec.get("relaxation") or 0.0),
                     natural_rows=0 if compat else 1)
        h = ctypes.c_void_p()
        rc = self._lib.fsor_create(ctypes.byref(cs), ctypes.byref(h))
        if rc != 0:
            raise FusionPicError(rc, self._lib.fsor_last_error(None).decode())
        self._h = h
        L, vh = ctypes.c_uint64(), ctypes.c_uint32()
        self._check(self._lib.fsor_dims(self._h, ctypes.byref(L), ctypes.byref(vh)))
        self.vec_length, self.vec_height = int(L.value), int(vh.value)     # out.vec_length, out.vec_height (:50-51)

    def _check(self, rc):
        if rc != 0:
            raise FusionPicError(rc, self._lib.fsor_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fsor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the reference's methods
    def set_matrix(self, matrix):
        """row-major 2-D array, matrix[row][col] (matrix_webgl.js:456-475)"""
        a, code = _host(matrix)
        if a.size != self.vec_length ** 2:
            raise FusionPicError(-1, "matrix must hold vec_length^2 = %d elements" % self.vec_length ** 2)
        self._check(self._lib.fsor_set_matrix(self._h, a.ctypes.data, code))
        return self

    def set_b(self, b):
        a, code = _host(b)
        if a.size != self.vec_length:
            raise FusionPicError(-1, "b must hold vec_length = %d elements" % self.vec_length)
        self._check(self._lib.fsor_set_b(self._h, a.ctypes.data, code))
        return self

    def init_vector(self, vector):
        a, code = _host(vector)
        if a.size != self.vec_length:
            raise FusionPicError(-1, "vector must hold vec_length = %d elements" % self.vec_length)
        self._check(self._lib.fsor_init_vector(self._h, a.ctypes.data, code))
        return self

    def mv_product(self, target=None):
        """x_guess <- x_result; x_result <- R x_guess + C (:535-558).  `target` is accepted for
        signature compatibility; the product always lands in x_result."""
        self._check(self._lib.fsor_iterate(self._h, 1))
        return self

    def solve(self, params):
        _validate(params, (("tolerance", False), ("substep", True), ("max_iterations", True)))
        res = SorResult()
        out = np.empty(self.vec_length, dtype=np.float32)
        has_max = params.get("max_iterations") is not None
        self._check(self._lib.fsor_solve(self._h, float(params["tolerance"]), int(params.get("substep") or 0), int(has_max),
                                         int(params.get("max_iterations") or 0), ctypes.byref(res), out.ctypes.data))
        return {"correlation": res.correlation, "diff": res.diff, "iterations": res.iterations, "result": out}

    def x_result_tex(self):
        """(device pointer, bytes) of x_result, the analogue of the frame buffer (:701-704)"""
        return self.deviceBuffer(X_RESULT)

    # ---- extensions
    def prepare(self):
        self._check(self._lib.fsor_prepare(self._h))
        return self

    def iterate(self, n=1):
        self._check(self._lib.fsor_iterate(self._h, int(n)))
        return self

    def readVector(self, which=X_RESULT):
        out = np.empty(self.vec_length, dtype=np.float32)
        self._check(self._lib.fsor_read_vector(self._h, which, out.ctypes.data))
        return out

    def readIterationMatrix(self):
        out = np.empty(self.vec_length ** 2, dtype=np.float32)
        self._check(self._lib.fsor_read_iteration_matrix(self._h, out.ctypes.data))
        return out

    def deviceBuffer(self, which=X_RESULT):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._lib.fsor_device_buffer(self._h, which, ctypes.byref(p), ctypes.byref(n)))
        return int(p.value), int(n.value)

    def setStream(self, stream):
        self._check(self._lib.fsor_set_stream(self._h, ctypes.c_void_p(stream)))

    def sync(self):
        self._check(self._lib.fsor_sync(self._h))

    def profile(self, on=True):
        self._check(self._lib.fsor_profile(self._h, int(bool(on))))

    def stats(self):
        st = SorStats()
        self._check(self._lib.fsor_get_stats(self._h, ctypes.byref(st)))
        return {"iterations": st.iterations, "seconds_iterate": st.seconds_iterate, "matrix_bytes": st.matrix_bytes}

    def resetStats(self):
        self._check(self._lib.fsor_reset_stats(self._h))


def makeSORIterative(spec, device=0, compat=True):
    return SORIterative(spec, device=device, compat=compat)
