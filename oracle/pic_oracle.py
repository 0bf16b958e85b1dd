"""ctypes front-end of the CPU oracle (oracle/libpic_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product (fusion-sim_amd/).

`OracleSim` mirrors the object the reference factory returns
(empic.js:30-1529: set / add* / precalc / step / density) on top of the C
restatement, in float32 (the reference's precision) or float64.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpic_oracle.so")

N_ENTROPY = 1024
N_CDF = 512


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src_m = max(os.path.getmtime(os.path.join(_HERE, f))
                for f in ("pic_oracle.c", "pic_oracle_impl.h", "pic_oracle.h", "sor_oracle.c", "es3d_oracle.c", "es3d_oracle_impl.h"))
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < src_m:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpic_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_lib_omp = None


def lib_omp():
    """libpic_oracle_omp.so: the same sources built with -fopenmp (per-particle loops in parallel,
    threaded deposit).  Used only by bench.py's all-cores CPU baseline."""
    global _lib_omp
    if _lib_omp is None:
        path = os.path.join(_HERE, "libpic_oracle_omp.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libpic_oracle_omp.so"], stdout=subprocess.DEVNULL)
        _lib_omp = ctypes.CDLL(path)
        _lib_omp.orc_tofixed20.restype = ctypes.c_double
        _lib_omp.orc_tofixed20.argtypes = [ctypes.c_double]
    return _lib_omp


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("PIC_ORACLE_LIB")   # `make -C oracle sanitize`: an ASan/UBSan build
        if not override:
            build()
        _lib = ctypes.CDLL(override or _LIB_PATH)
        _lib.orc_tofixed20.restype = ctypes.c_double
        _lib.orc_tofixed20.argtypes = [ctypes.c_double]
        _lib.orc_inv_cdf.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _real(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "orc_f32_", ctypes.c_float
    if dtype == np.float64:
        return "orc_f64_", ctypes.c_double
    raise ValueError("dtype must be float32 or float64")


def constants(spec):
    out = (ctypes.c_double * 6)()
    lib().orc_constants(ctypes.c_double(spec["radius"]), ctypes.c_double(spec["height"]), ctypes.c_double(spec["dt"]),
                        ctypes.c_double(spec["particle_mass"]), ctypes.c_double(spec["particle_charge"]), out)
    return dict(h=out[0], factor_r=out[1], factor_z=out[2], step_factor=out[3], f_rz=out[4], f_zr=out[5])


def tofixed20(x):
    return lib().orc_tofixed20(float(x))


def stamp():
    out = np.zeros(121, dtype=np.float32)
    lib().orc_stamp(_p(out))
    return out


def inv_cdf(pdf):
    """pdf[nr][nz] -> (rc, table float32[512*512*4])."""
    pdf = np.ascontiguousarray(pdf, dtype=np.float64)
    out = np.zeros(4 * N_CDF * N_CDF, dtype=np.float32)
    rc = lib().orc_inv_cdf(_p(pdf), ctypes.c_int(pdf.shape[0]), ctypes.c_int(pdf.shape[1]), _p(out))
    return rc, out


class OracleSim:
    """CPU twin of empic.makeCylindricalParticlePusher(spec) (empic.js:30)."""

    def __init__(self, spec, dtype=np.float32, physical_a=False, count=None, rng="reference", seed=0, threads=1, shape="ref11", raster_bits=0):
        # raster_bits > 0: the point sprites as a rasteriser with that many sub-pixel bits draws them (deposit_raster in
        # pic_oracle_impl.h: snapped window coordinates, y down, cropped instead of discarded); 0: the ideal sprite
        self.raster_bits = int(raster_bits)
        self.shape = shape  # "ref11": the reference's 11x11 stamp; "cic": the bilinear extension
        # threads > 1: the OpenMP build (timing only: the threaded deposit sums in another order)
        self.threads = int(threads)
        self._lib = lib_omp() if self.threads > 1 else lib()
        if self.threads > 1:
            self._lib.orc_set_threads(self.threads)
        # rng="counter": the Philox extension mode (no per-particle random state), sub-step counter self.t
        self.rng, self.seed, self.t = rng, int(seed), 0
        self.spec = dict(spec)
        self.dtype = np.dtype(dtype)
        self.prefix, self.creal = _real(dtype)
        self.physical_a = int(bool(physical_a))
        self.nr, self.nz = int(spec["nr"]), int(spec["nz"])
        self.n = int(count) if count else int(spec["nparticles"]) ** 2
        self.k = constants(spec)
        T = self.dtype.type
        if self.dtype == np.float32:
            # uniforms are uploaded with uniform1f, literals re-read by the GLSL compiler
            self.h = T(self.k["h"])
            self.step_factor = T(self.k["step_factor"])
            self.f_rz = T(tofixed20(self.k["f_rz"]))
            self.f_zr = T(tofixed20(self.k["f_zr"]))
            self.fr = T(tofixed20(self.k["factor_r"]))
            self.fz = T(tofixed20(self.k["factor_z"]))
        else:
            self.h, self.step_factor = T(self.k["h"]), T(self.k["step_factor"])
            self.f_rz, self.f_zr = T(self.k["f_rz"]), T(self.k["f_zr"])
            self.fr, self.fz = T(self.k["factor_r"]), T(self.k["factor_z"])
        nc = self.nr * self.nz
        z4 = lambda m: np.zeros(4 * m, dtype=self.dtype)
        self.pos_A, self.vel_A, self.rand_A = z4(self.n), z4(self.n), z4(self.n)
        self.pos_B, self.vel_B, self.rand_B = z4(self.n), z4(self.n), z4(self.n)
        self.entropy = z4(N_ENTROPY * N_ENTROPY)
        self.E, self.B, self.sink = z4(nc), z4(nc), z4(nc)
        self.inv_cdf = z4(N_CDF * N_CDF)
        self.R1, self.R2, self.R3, self.A = z4(nc), z4(nc), z4(nc), z4(nc)
        self.moments, self.norm, self.avg_A, self.avg_B = z4(nc), z4(nc), z4(nc), z4(nc)
        self.stamp = stamp()
        self._shapes = None

    def _f(self, name):
        return getattr(self._lib, self.prefix + name)

    def _c(self, v):
        return self.creal(float(v))

    # ---- out.set (empic.js:1157-1350)
    def set(self, position=None, velocity=None, E=None, B=None, sink_mask=None, source_pdf=None):
        if E is not None:
            self._f("pack_grid3")(_p(np.ascontiguousarray(E, dtype=np.float64)), self.nr, self.nz, _p(self.E))
        if B is not None:
            self._f("pack_grid3")(_p(np.ascontiguousarray(B, dtype=np.float64)), self.nr, self.nz, _p(self.B))
        if position is not None:
            a = np.ascontiguousarray(position, dtype=np.float64)
            assert a.shape == (self.n, 3)
            self._f("normalise_particles")(_p(a), ctypes.c_size_t(self.n), ctypes.c_double(self.k["factor_r"]),
                                           ctypes.c_double(self.k["factor_z"]), _p(self.pos_A))
            self.pos_B[:] = self.pos_A
        if velocity is not None:
            a = np.ascontiguousarray(velocity, dtype=np.float64)
            assert a.shape == (self.n, 3)
            self._f("normalise_particles")(_p(a), ctypes.c_size_t(self.n), ctypes.c_double(self.k["factor_r"]),
                                           ctypes.c_double(self.k["factor_z"]), _p(self.vel_A))
            self.vel_B[:] = self.vel_A
        if sink_mask is not None:
            self._f("pack_sink")(_p(np.ascontiguousarray(sink_mask, dtype=np.float64)), self.nr, self.nz, _p(self.sink))
        if source_pdf is not None:
            rc, tab = inv_cdf(source_pdf)
            if rc != 0:
                raise TypeError("reference set({source_pdf}) throws for this pdf")
            self.inv_cdf[:] = tab.astype(self.dtype)

    def set_random_state(self, entropy=None, rand=None):
        if entropy is not None:
            self.entropy[:] = np.asarray(entropy, dtype=np.float32).ravel().astype(self.dtype)
        if rand is not None:
            self.rand_A[:] = np.asarray(rand, dtype=np.float32).ravel().astype(self.dtype)

    # ---- painters (empic.js:1352-1411)
    def add_bz(self, bz):
        self._f("add_uniform")(_p(self.B), self.nr, self.nz, 1, self._c(self.dtype.type(bz)))

    def add_btheta(self, bt):
        self._f("add_uniform")(_p(self.B), self.nr, self.nz, 2, self._c(self.dtype.type(bt)))

    def add_current_z(self, current):
        self._f("add_uniform")(_p(self.B), self.nr, self.nz, 0, self._c(self.dtype.type(current)))

    def add_current_loop(self, r, z, current):
        if self._shapes is None:
            half = np.zeros(4 * self.nr * self.nz, dtype=self.dtype)
            tenth = np.zeros_like(half)
            self._f("loop_shape")(self._c(0.5), self.nr, self.nz, _p(half))
            self._f("loop_shape")(self._c(0.1), self.nr, self.nz, _p(tenth))
            self._shapes = (half, tenth)
        T = self.dtype.type
        self._f("add_current_loop")(_p(self.B), _p(self._shapes[0]), _p(self._shapes[1]), self.nr, self.nz,
                                    self._c(T(r * self.k["factor_r"])), self._c(T(z * self.k["factor_z"])),
                                    self._c(T(current)))

    # ---- out.precalc (empic.js:1413-1434)
    def precalc(self):
        self._f("precalc")(_p(self.B), _p(self.E), self.nr, self.nz, self._c(self.h), self._c(self.fr), self._c(self.fz),
                           self._c(self.f_rz), self._c(self.f_zr), _p(self.R1), _p(self.R2), _p(self.R3), _p(self.A),
                           self.physical_a)

    # ---- out.step (empic.js:1436-1469)
    def step(self, ncalls=1):
        if self.rng == "counter":
            self._f("step_counter")(_p(self.pos_A), _p(self.vel_A), _p(self.pos_B), _p(self.vel_B), _p(self.R1), _p(self.R2),
                                    _p(self.R3), _p(self.A), _p(self.sink), _p(self.inv_cdf), self.nr, self.nz,
                                    self._c(self.step_factor), ctypes.c_size_t(self.n), int(ncalls),
                                    ctypes.c_uint64(self.seed), ctypes.c_uint64(self.t))
            self.t += 2 * int(ncalls)
            return
        self._f("step")(_p(self.pos_A), _p(self.vel_A), _p(self.rand_A), _p(self.pos_B), _p(self.vel_B), _p(self.rand_B),
                        _p(self.entropy), _p(self.R1), _p(self.R2), _p(self.R3), _p(self.A), _p(self.sink),
                        _p(self.inv_cdf), self.nr, self.nz, self._c(self.step_factor), ctypes.c_size_t(self.n),
                        int(ncalls))

    # ---- out.density (empic.js:1471-1495)
    def deposit(self):
        if self.shape == "cic":   # extension: bilinear deposit on the four nearest cell centres
            self._f("deposit_cic")(_p(self.pos_A), _p(self.vel_A), ctypes.c_size_t(self.n), self.nr, self.nz, _p(self.moments))
            return
        if self.raster_bits:
            self._f("deposit_raster")(_p(self.pos_A), _p(self.vel_A), ctypes.c_size_t(self.n), _p(self.stamp), self.nr, self.nz,
                                      _p(self.moments), self.raster_bits)
            return
        self._f("deposit_threads" if self.threads > 1 else "deposit")(
            _p(self.pos_A), _p(self.vel_A), ctypes.c_size_t(self.n), _p(self.stamp), self.nr, self.nz, _p(self.moments))

    def density_finish(self):
        self._f("normalise")(_p(self.moments), self.nr, self.nz, _p(self.norm))
        self._f("avg")(_p(self.norm), _p(self.avg_B), _p(self.avg_A), self._c(self.dtype.type(0.01)),
                       ctypes.c_size_t(self.nr * self.nz))

    def density(self):
        self.deposit()
        self.density_finish()

    # ---- read-back helpers
    def cells(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._f("cells")(_p(self.pos_A), ctypes.c_size_t(self.n), self.nr, self.nz, _p(out))
        return out

    def deposit_cells(self):
        out = np.zeros(self.n, dtype=np.int32)
        self._f("deposit_cells")(_p(self.pos_A), ctypes.c_size_t(self.n), self.nr, self.nz, _p(out))
        return out

    def raster_cells(self, bits=None):
        """Sprite-centre cells (ci, cj) under the rasterised convention; INT32_MIN for dropped points."""
        ci, cj = np.zeros(self.n, dtype=np.int32), np.zeros(self.n, dtype=np.int32)
        self._f("raster_cells")(_p(self.pos_A), ctypes.c_size_t(self.n), self.nr, self.nz, int(bits or self.raster_bits), _p(ci), _p(cj))
        return ci, cj

    def positions(self):
        return self.pos_A.reshape(self.n, 4)[:, :3]

    def velocities(self):
        return self.vel_A.reshape(self.n, 4)[:, :3]

    def alive(self):
        return (self.pos_A.reshape(self.n, 4)[:, 3] > 0.5).astype(np.uint8)

    def rand(self):
        return self.rand_A.reshape(self.n, 4)
