"""fusionpic — Python host mirror of the reference's pusher object over the C ABI.

The shipped host language is JavaScript (fusion-sim_amd/js/empic_native.js over an
N-API addon).  This module binds the SAME C ABI (include/fusionpic.h) with ctypes
for the Python tools of this repository: tests/, bench.py and __graft_entry__.py.
It keeps the reference's factory and method names
(empic.js:30 makeCylindricalParticlePusher, :1157 set, :1352 addCurrentLoop,
:1380 addCurrentZ, :1391 addBZ, :1402 addBTheta, :1413 precalc, :1436 step,
:1471 density) and its error behaviour: a failed call raises `FusionPicError`
synchronously, with spec errors worded ".prop <- ..." (utilities.js:118-127).

There is no CPU path here.  If libfusionpic.so is missing the import fails; if no
gfx950 device is present the factory raises.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libfusionpic.so")

F32, F64 = 0, 1
GRID_E, GRID_B, GRID_SINK_MASK, GRID_SOURCE_PDF = 0, 1, 2, 3
(READ_MOMENTS, READ_NORM, READ_AVG, READ_R1, READ_R2, READ_R3, READ_A, READ_B, READ_E, READ_SINK,
 READ_INV_CDF) = range(11)
BUF_CELL_SUMS, BUF_RHO_FIXED = 0, 1
GEOM_CYL_RZ, GEOM_CART3D = 0, 1
SOLVER_NONE, SOLVER_POISSON_FFT, SOLVER_YEE = 0, 1, 2
F3_E, F3_RHO, F3_PHI, F3_RHO_FIXED, F3_B_NODES, F3_EDGE_E, F3_FACE_B, F3_J_FIXED = range(8)

ABI_FUNCTIONS = [
    "fpic_last_error", "fpic_abi_version", "fpic_build_arch", "fpic_create", "fpic_destroy", "fpic_set_particles",
    "fpic_set_grid", "fpic_set_random_state", "fpic_add_current_loop", "fpic_add_current_z", "fpic_add_bz",
    "fpic_add_btheta", "fpic_precalc", "fpic_step", "fpic_substeps", "fpic_density", "fpic_deposit", "fpic_density_finish",
    "fpic_density_finish_from",
    "fpic_read_grid", "fpic_get_particles", "fpic_get_cells", "fpic_device_buffer", "fpic_set_stream",
    "fpic_get_stream", "fpic_sort", "fpic_sync", "fpic_profile", "fpic_get_stats", "fpic_reset_stats",
    "fpic_get_substep_counter", "fpic_set_substep_counter", "fpic_save_checkpoint", "fpic_load_checkpoint",
    "fpic_add_species", "fpic_set_particles_of", "fpic_get_particles_of", "fpic_get_cells_of", "fpic_add_b",
    "fpic_set_field3", "fpic_read_field3", "fpic_set_particles_range", "fpic_get_particles_range", "fpic_get_cells_range",
    "fpic_comm_unique_id", "fpic_comm_init", "fpic_comm_destroy", "fpic_comm_info", "fpic_comm_set_overlap",
    "fpic_domain_init", "fpic_domain_set_particles", "fpic_domain_get_particles", "fpic_domain_stats",
    "fpic_group_precalc", "fpic_group_step", "fpic_group_density",
]


class FusionPicError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


class Spec(ctypes.Structure):
    _fields_ = [
        ("radius", ctypes.c_double), ("height", ctypes.c_double), ("nr", ctypes.c_int32), ("nz", ctypes.c_int32),
        ("dt", ctypes.c_double), ("nparticles", ctypes.c_int32), ("particle_mass", ctypes.c_double),
        ("particle_charge", ctypes.c_double), ("count", ctypes.c_uint64), ("precision", ctypes.c_int32),
        ("device", ctypes.c_int32), ("physical_a", ctypes.c_int32), ("sort_interval", ctypes.c_int32),
        ("unfused_deposit", ctypes.c_int32), ("rng_mode", ctypes.c_int32), ("rng_seed_lo", ctypes.c_uint32),
        ("rng_seed_hi", ctypes.c_uint32), ("geometry", ctypes.c_int32), ("solver", ctypes.c_int32), ("ny", ctypes.c_int32),
        ("shape", ctypes.c_int32), ("length_y", ctypes.c_double), ("macro_weight", ctypes.c_double),
        ("raster_subpixel_bits", ctypes.c_int32), ("reserved_i32", ctypes.c_int32), ("reserved", ctypes.c_double * 5),
    ]


class Stats(ctypes.Structure):
    _fields_ = [
        ("n_particles", ctypes.c_uint64), ("particle_updates", ctypes.c_uint64), ("step_launches", ctypes.c_uint64),
        ("deposit_launches", ctypes.c_uint64), ("sort_passes", ctypes.c_uint64), ("deposit_spilled", ctypes.c_uint64),
        ("ms_push", ctypes.c_double), ("ms_deposit", ctypes.c_double), ("ms_stamp", ctypes.c_double),
        ("ms_precalc", ctypes.c_double), ("ms_sort", ctypes.c_double), ("bytes_particle_state", ctypes.c_uint64),
        ("bytes_grid_state", ctypes.c_uint64), ("ms_solve", ctypes.c_double), ("solve_launches", ctypes.c_uint64),
        ("reserved", ctypes.c_double * 6),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


_lib = None


def load_library(path=None):
    """dlopen libfusionpic.so.  Fails loudly when the HIP library has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # PyTorch-ROCm ships its own copy of the HIP runtime.  Two runtimes in one process do not share
    # the device (the second one reports "No HIP GPUs"), so when torch is installed it is imported
    # first and libfusionpic.so then binds to the runtime already loaded.  FUSIONPIC_NO_TORCH=1 skips
    # this for hosts that never touch torch.
    if "torch" not in sys.modules and not os.environ.get("FUSIONPIC_NO_TORCH"):
        import importlib.util
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `make -C fusion-sim_amd` (hipcc, gfx950); "
                          "there is no CPU fallback" % path)
    lib = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.fpic_last_error.restype = ctypes.c_char_p
    lib.fpic_last_error.argtypes = [vp]
    lib.fpic_build_arch.restype = ctypes.c_char_p
    lib.fpic_create.argtypes = [ctypes.POINTER(Spec), ctypes.POINTER(vp)]
    lib.fpic_destroy.argtypes = [vp]
    lib.fpic_set_particles.argtypes = [vp, vp, vp, ctypes.c_uint64, ci]
    lib.fpic_set_grid.argtypes = [vp, ci, vp, ci, ci, ci, ci]
    lib.fpic_set_random_state.argtypes = [vp, vp, vp]
    lib.fpic_add_current_loop.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    for f in ("fpic_add_current_z", "fpic_add_bz", "fpic_add_btheta"):
        getattr(lib, f).argtypes = [vp, ctypes.c_double]
    for f in ("fpic_precalc", "fpic_density", "fpic_deposit", "fpic_density_finish", "fpic_sort", "fpic_sync",
              "fpic_reset_stats"):
        getattr(lib, f).argtypes = [vp]
    lib.fpic_density_finish_from.argtypes = [vp, vp, vp]
    lib.fpic_step.argtypes = [vp, ci]
    lib.fpic_substeps.argtypes = [vp, ci]
    lib.fpic_profile.argtypes = [vp, ci]
    lib.fpic_read_grid.argtypes = [vp, ci, vp, ci]
    lib.fpic_get_particles.argtypes = [vp, vp, vp, vp, vp, ci]
    lib.fpic_get_cells.argtypes = [vp, vp]
    lib.fpic_device_buffer.argtypes = [vp, ci, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    lib.fpic_set_stream.argtypes = [vp, vp]
    lib.fpic_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.fpic_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.fpic_get_substep_counter.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.fpic_set_substep_counter.argtypes = [vp, ctypes.c_uint64]
    lib.fpic_save_checkpoint.argtypes = [vp, ctypes.c_char_p]
    lib.fpic_load_checkpoint.argtypes = [vp, ctypes.c_char_p]
    lib.fpic_comm_unique_id.argtypes = [vp]
    lib.fpic_comm_init.argtypes = [vp, vp, ci, ci]
    lib.fpic_comm_destroy.argtypes = [vp]
    lib.fpic_comm_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.fpic_comm_set_overlap.argtypes = [vp, ci]
    lib.fpic_domain_init.argtypes = [vp, ci, ci, ci, ci, ci]
    lib.fpic_domain_set_particles.argtypes = [vp, ci, ctypes.c_uint64, vp, vp, ctypes.c_uint32, ci]
    lib.fpic_domain_get_particles.argtypes = [vp, ci, vp, vp, vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ci]
    lib.fpic_domain_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.fpic_group_precalc.argtypes = [ctypes.POINTER(vp), ci]
    lib.fpic_group_step.argtypes = [ctypes.POINTER(vp), ci, ci]
    lib.fpic_group_density.argtypes = [ctypes.POINTER(vp), ci]
    lib.fpic_add_species.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_uint64, ctypes.POINTER(ci)]
    lib.fpic_set_particles_of.argtypes = [vp, ci, vp, vp, ctypes.c_uint64, ci]
    lib.fpic_get_particles_of.argtypes = [vp, ci, vp, vp, ci]
    lib.fpic_set_particles_range.argtypes = [vp, ci, ctypes.c_uint64, ctypes.c_uint64, vp, vp, ci]
    lib.fpic_get_cells_of.argtypes = [vp, ci, vp]
    lib.fpic_get_particles_range.argtypes = [vp, ci, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, vp, vp, ci]
    lib.fpic_get_cells_range.argtypes = [vp, ci, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, vp]
    lib.fpic_add_b.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.fpic_set_field3.argtypes = [vp, ci, vp, ci, ci, ci, ci]
    lib.fpic_read_field3.argtypes = [vp, ci, vp, ci]
    if path == LIB_PATH:
        _lib = lib
    return lib


_SPEC_KEYS = ("radius", "height", "nr", "nz", "dt", "nparticles", "particle_mass", "particle_charge")


def _validate_spec(spec):
    """validate_object(spec, {...: 'number'}) (empic.js:31-41, utilities.js:11-127)."""
    for key in _SPEC_KEYS:
        if key not in spec or spec[key] is None:
            raise FusionPicError(-1, "." + key + " <- Non-optional property is undefined!")
        if isinstance(spec[key], bool) or not isinstance(spec[key], (int, float, np.integer, np.floating)):
            raise FusionPicError(-1, "." + key + " <- Property does not match any given possible types!")


def _np_dtype(code):
    return np.float32 if code == F32 else np.float64


def _code(arr):
    return F32 if arr.dtype == np.float32 else F64


def _as_float_array(a):
    a = np.asarray(a)
    if a.dtype != np.float32:
        a = a.astype(np.float64, copy=False)  # JavaScript numbers are doubles
    return np.ascontiguousarray(a)


class _Buffer:
    """pointer + shape + element code of an [n][3] array the library reads: a numpy array, or device memory
    (a tensor with data_ptr(), e.g. torch on the GPU) for the entry points that copy with hipMemcpyDefault"""

    def __init__(self, ptr, shape, code, keep):
        self.ptr, self.shape, self.code, self.keep = ptr, tuple(shape), code, keep


def _device_or_host(a):
    if hasattr(a, "data_ptr") and hasattr(a, "is_contiguous"):
        name = str(a.dtype)
        if not a.is_contiguous() or not (name.endswith("float32") or name.endswith("float64")):
            raise FusionPicError(-1, ".position <- a device tensor must be contiguous float32 or float64")
        return _Buffer(a.data_ptr(), a.shape, F32 if name.endswith("float32") else F64, a)
    h = _as_float_array(a)
    return _Buffer(h.ctypes.data, h.shape, _code(h), h)


class CylindricalParticlePusher:
    """Object returned by makeCylindricalParticlePusher (empic.js:1528)."""

    def __init__(self, spec, precision="fp32", device=0, count=0, compat=True, sort_interval=0, fuse_deposit=True,
                 rng="reference", seed=0, library=None, shape="ref11", raster_subpixel_bits=0):
        _validate_spec(spec)
        self._lib = library or load_library()
        self.spec = dict(spec)
        s = Spec()
        for key in _SPEC_KEYS:
            setattr(s, key, spec[key])
        s.count = int(count)
        s.precision = {"fp32": F32, "fp64": F64}[precision]
        s.device = int(device)
        s.physical_a = 0 if compat else 1
        s.sort_interval = int(sort_interval)
        # True: sums, census and re-binning inside the push; False: separate passes; "census": the push
        # keeps the census and the re-binning, the per-cell sums are a separate pass
        s.unfused_deposit = 2 if fuse_deposit == "census" else (0 if fuse_deposit else 1)
        s.rng_mode = {"reference": 0, "counter": 1}[rng]
        s.shape = {"ref11": 0, "cic": 1}[spec.get("shape", shape)]   # SURVEY 8(b) extension key
        # density()'s point sprites as a rasteriser with that many sub-pixel bits draws them (0 / absent: ideal sprites)
        s.raster_subpixel_bits = int(spec.get("raster_subpixel_bits", raster_subpixel_bits))
        s.rng_seed_lo, s.rng_seed_hi = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
        self.precision = s.precision
        self.nr, self.nz = int(spec["nr"]), int(spec["nz"])
        self.n = int(count) if count else int(spec["nparticles"]) ** 2
        h = ctypes.c_void_p()
        rc = self._lib.fpic_create(ctypes.byref(s), ctypes.byref(h))
        if rc != 0:
            raise FusionPicError(rc, self._lib.fpic_last_error(None).decode())
        self._h = h

    # ---- plumbing
    def _check(self, rc):
        if rc != 0:
            raise FusionPicError(rc, self._lib.fpic_last_error(self._h).decode())

    def destroy(self):
        if getattr(self, "_h", None):
            self._lib.fpic_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # ---- reference surface
    def set(self, value=None, **kw):
        """out.set({E, B, position, velocity, sink_mask, source_pdf}) (empic.js:1157-1350)."""
        value = dict(value or {}, **kw)
        for key, which, ncomp in (("E", GRID_E, 3), ("B", GRID_B, 3)):
            if value.get(key) is not None:
                a = _as_float_array(value[key])
                if a.shape != (self.nr, self.nz, 3):
                    raise FusionPicError(-1, ".%s <- expected [%d][%d][3]" % (key, self.nr, self.nz))
                self._check(self._lib.fpic_set_grid(self._h, which, a.ctypes.data, self.nr, self.nz, ncomp, _code(a)))
        pos = value.get("position")
        vel = value.get("velocity")
        for key, arr in (("position", pos), ("velocity", vel)):
            if arr is not None:
                a = _as_float_array(arr)
                if a.shape != (self.n, 3):
                    raise FusionPicError(-1, ".%s <- expected [%d][3]" % (key, self.n))
                p = a.ctypes.data if key == "position" else None
                v = a.ctypes.data if key == "velocity" else None
                self._check(self._lib.fpic_set_particles(self._h, p, v, self.n, _code(a)))
        for key, which in (("sink_mask", GRID_SINK_MASK), ("source_pdf", GRID_SOURCE_PDF)):
            if value.get(key) is not None:
                a = _as_float_array(value[key])
                if a.shape != (self.nr, self.nz):
                    raise FusionPicError(-1, ".%s <- expected [%d][%d]" % (key, self.nr, self.nz))
                self._check(self._lib.fpic_set_grid(self._h, which, a.ctypes.data, self.nr, self.nz, 1, _code(a)))

    def addCurrentLoop(self, r, z, current):
        self._check(self._lib.fpic_add_current_loop(self._h, r, z, current))

    def addCurrentZ(self, current):
        self._check(self._lib.fpic_add_current_z(self._h, current))

    def addBZ(self, bz):
        self._check(self._lib.fpic_add_bz(self._h, bz))

    def addBTheta(self, btheta):
        self._check(self._lib.fpic_add_btheta(self._h, btheta))

    def addSpindleCuspPlasmaField(self, r, B_c, beta_c=None):
        """The reference's version is unfinished and has no effect on B (empic.js:1369-1378,
        spindle.js:328, :624, :643 use undefined symbols)."""
        raise FusionPicError(-5, "addSpindleCuspPlasmaField is not functional in the reference (spindle.js:328)")

    def precalc(self):
        self._check(self._lib.fpic_precalc(self._h))

    def step(self, ncalls=1):
        """One call = two leap-frog sub-steps, dt fixed at construction (empic.js:1436-1469)."""
        self._check(self._lib.fpic_step(self._h, int(ncalls)))

    def substeps(self, nsub):
        """nsub single leap-frog sub-steps: step(n) == substeps(2 n) (diagnostics that need the state between the halves)"""
        self._check(self._lib.fpic_substeps(self._h, int(nsub)))

    def density(self):
        self._check(self._lib.fpic_density(self._h))

    # ---- extensions the boundary needs because `canvas` cannot exist off-browser
    def deposit(self):
        self._check(self._lib.fpic_deposit(self._h))

    def densityFinish(self):
        self._check(self._lib.fpic_density_finish(self._h))

    def densityFinishFrom(self, sums_ptr, stream=None):
        """finish stage from a caller's copy of the per-cell sums, on a caller's stream (multi-GPU overlap)"""
        self._check(self._lib.fpic_density_finish_from(self._h, ctypes.c_void_p(sums_ptr), ctypes.c_void_p(stream or 0)))

    def setRandomState(self, entropy=None, rand=None):
        e = r = None
        if entropy is not None:
            e = np.ascontiguousarray(entropy, dtype=np.float32).ravel()
            if e.size != 4 * 1024 * 1024:
                raise FusionPicError(-1, ".entropy <- expected 1024*1024*4 floats")
        if rand is not None:
            r = np.ascontiguousarray(rand, dtype=np.float32).ravel()
            if r.size != 4 * self.n:
                raise FusionPicError(-1, ".rand <- expected %d*4 floats" % self.n)
        self._check(self._lib.fpic_set_random_state(self._h, e.ctypes.data if e is not None else None,
                                                    r.ctypes.data if r is not None else None))

    def readGrid(self, which, dtype=None):
        code = self.precision if dtype is None else (F32 if np.dtype(dtype) == np.float32 else F64)
        cells = 512 * 512 if which == READ_INV_CDF else self.nr * self.nz
        out = np.empty(4 * cells, dtype=_np_dtype(code))
        self._check(self._lib.fpic_read_grid(self._h, which, out.ctypes.data, code))
        return out

    def readDensity(self, dtype=None):
        """moments01_avgA, channels (v_r, v_theta, v_z, n), index 4*(i + j*nr) (empic.js:1071)."""
        return self.readGrid(READ_AVG, dtype)

    def readMoments(self, dtype=None):
        return self.readGrid(READ_MOMENTS, dtype)

    def getParticles(self, dtype=None, position=True, velocity=True, rand=True, alive=True):
        code = self.precision if dtype is None else (F32 if np.dtype(dtype) == np.float32 else F64)
        dt = _np_dtype(code)
        out = {}
        if position:
            out["position"] = np.empty((self.n, 3), dtype=dt)
        if velocity:
            out["velocity"] = np.empty((self.n, 3), dtype=dt)
        if rand:
            out["rand"] = np.empty((self.n, 4), dtype=np.float32)
        if alive:
            out["alive"] = np.empty(self.n, dtype=np.uint8)
        ptr = lambda k: out[k].ctypes.data if k in out else None
        self._check(self._lib.fpic_get_particles(self._h, ptr("position"), ptr("velocity"), ptr("rand"), ptr("alive"), code))
        return out

    def getCells(self):
        out = np.empty(self.n, dtype=np.int32)
        self._check(self._lib.fpic_get_cells(self._h, out.ctypes.data))
        return out

    def deviceBuffer(self, which=BUF_CELL_SUMS):
        p, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._lib.fpic_device_buffer(self._h, which, ctypes.byref(p), ctypes.byref(nbytes)))
        return p.value, nbytes.value

    def setStream(self, stream_ptr):
        self._check(self._lib.fpic_set_stream(self._h, ctypes.c_void_p(stream_ptr)))

    def substepCounter(self):
        t = ctypes.c_uint64()
        self._check(self._lib.fpic_get_substep_counter(self._h, ctypes.byref(t)))
        return t.value

    def setSubstepCounter(self, t):
        self._check(self._lib.fpic_set_substep_counter(self._h, int(t)))

    def saveCheckpoint(self, path):
        self._check(self._lib.fpic_save_checkpoint(self._h, os.fsencode(path)))

    def loadCheckpoint(self, path):
        self._check(self._lib.fpic_load_checkpoint(self._h, os.fsencode(path)))

    def sort(self):
        self._check(self._lib.fpic_sort(self._h))

    # ---- multi-GPU: the library's own RCCL communicator (include/fusionpic.h, fpic_comm_*)
    def commInit(self, unique_id, rank, world, overlap=True):
        """unique_id: the 128 bytes rank 0 obtained from commUniqueId(), handed to every rank by the host"""
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.fpic_comm_init(self._h, buf, int(rank), int(world)))
        self._check(self._lib.fpic_comm_set_overlap(self._h, 1 if overlap else 0))

    def commDestroy(self):
        self._check(self._lib.fpic_comm_destroy(self._h))

    def commInfo(self):
        r, w = ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.fpic_comm_info(self._h, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    def sync(self):
        self._check(self._lib.fpic_sync(self._h))

    def profile(self, enable=True):
        self._check(self._lib.fpic_profile(self._h, 1 if enable else 0))

    def stats(self):
        s = Stats()
        self._check(self._lib.fpic_get_stats(self._h, ctypes.byref(s)))
        return s.as_dict()

    def resetStats(self):
        self._check(self._lib.fpic_reset_stats(self._h))


class ElectrostaticBoxPusher:
    """spec.geometry == 'cart3d': the self-consistent electrostatic extension (BASELINE.json
    configs[2..4]) behind the reference's method names.  Periodic box radius x length_y x height
    (x, y, z) on nr x ny x nz nodes; set / addBZ / precalc / step / density keep their meaning
    (include/fusionpic.h, "extension: spec.geometry").  No reference counterpart: parity unpinned."""

    def __init__(self, spec, precision="fp32", device=0, count=0, sort_interval=0, library=None, **ignored):
        _validate_spec(spec)
        for key in ("ny", "length_y"):
            if key not in spec or isinstance(spec[key], bool) or not isinstance(spec[key], (int, float, np.integer, np.floating)):
                raise FusionPicError(-1, "." + key + " <- Non-optional property is undefined!")
        self._lib = library or load_library()
        self.spec = dict(spec)
        s = Spec()
        for key in _SPEC_KEYS:
            setattr(s, key, spec[key])
        count = int(count or spec.get("count") or 0)
        s.count = count
        s.precision = {"fp32": F32, "fp64": F64}[spec.get("precision", precision)]
        s.device = int(device)
        s.sort_interval = int(sort_interval)
        s.geometry = GEOM_CART3D
        s.solver = {"none": SOLVER_NONE, "poisson_fft": SOLVER_POISSON_FFT, "yee": SOLVER_YEE}[spec.get("solver", "poisson_fft")]
        s.ny = int(spec["ny"])
        s.length_y = float(spec["length_y"])
        s.macro_weight = float(spec.get("macro_weight", 1.0))
        self.precision = s.precision
        self.nx, self.ny, self.nz = int(spec["nr"]), int(spec["ny"]), int(spec["nz"])
        self.nodes = self.nx * self.ny * self.nz
        self.counts = [count if count else int(spec["nparticles"]) ** 2]
        self.n = self.counts[0]
        h = ctypes.c_void_p()
        rc = self._lib.fpic_create(ctypes.byref(s), ctypes.byref(h))
        if rc != 0:
            raise FusionPicError(rc, self._lib.fpic_last_error(None).decode())
        self._h = h

    _check = CylindricalParticlePusher._check
    destroy = CylindricalParticlePusher.destroy
    __del__ = CylindricalParticlePusher.__del__
    sync = CylindricalParticlePusher.sync
    sort = CylindricalParticlePusher.sort
    profile = CylindricalParticlePusher.profile
    stats = CylindricalParticlePusher.stats
    resetStats = CylindricalParticlePusher.resetStats
    setStream = CylindricalParticlePusher.setStream
    precalc = CylindricalParticlePusher.precalc
    step = CylindricalParticlePusher.step
    substeps = CylindricalParticlePusher.substeps
    density = CylindricalParticlePusher.density
    addBZ = CylindricalParticlePusher.addBZ
    deviceBuffer = CylindricalParticlePusher.deviceBuffer
    commInit = CylindricalParticlePusher.commInit
    commDestroy = CylindricalParticlePusher.commDestroy
    commInfo = CylindricalParticlePusher.commInfo

    def addSpecies(self, mass, charge, count):
        idx = ctypes.c_int()
        self._check(self._lib.fpic_add_species(self._h, float(mass), float(charge), int(count), ctypes.byref(idx)))
        self.counts.append(int(count))
        return idx.value

    def addB(self, bx, by, bz):
        self._check(self._lib.fpic_add_b(self._h, float(bx), float(by), float(bz)))

    def set(self, value=None, species=0, **kw):
        """out.set({position, velocity, E}): positions in metres, velocities in units of c (empic.js:1199-1244);
        E is value[i][j][k][3] in V/m (a static field with solver 'none', or a field injected for a test)."""
        value = dict(value or {}, **kw)
        n = self.counts[species]
        for key in ("position", "velocity"):
            if value.get(key) is not None:
                a = _as_float_array(value[key])
                if a.shape != (n, 3):
                    raise FusionPicError(-1, ".%s <- expected [%d][3]" % (key, n))
                p = a.ctypes.data if key == "position" else None
                v = a.ctypes.data if key == "velocity" else None
                self._check(self._lib.fpic_set_particles_of(self._h, species, p, v, n, _code(a)))
        # E: node-centred (static field / injected field); edge_E, face_B: the Yee lattice's own arrays (full EM)
        for key, which in (("E", F3_E), ("edge_E", F3_EDGE_E), ("face_B", F3_FACE_B)):
            if value.get(key) is not None:
                a = _as_float_array(value[key])
                if a.shape != (self.nx, self.ny, self.nz, 3):
                    raise FusionPicError(-1, ".%s <- expected [%d][%d][%d][3]" % (key, self.nx, self.ny, self.nz))
                self._check(self._lib.fpic_set_field3(self._h, which, a.ctypes.data, self.nx, self.ny, self.nz, _code(a)))

    def setRange(self, first, position=None, velocity=None, species=0):
        """the caller's particles [first, first + len) of a species (piecewise upload of a large population); numpy
        arrays, or device-resident tensors (anything with data_ptr(): the library copies with hipMemcpyDefault, and
        the caller has synchronised the stream that produced them)"""
        arrs = [None if a is None else _device_or_host(a) for a in (position, velocity)]
        m = next(a.shape[0] for a in arrs if a is not None)
        codes = {a.code for a in arrs if a is not None}
        if len(codes) != 1 or any(a is not None and a.shape != (m, 3) for a in arrs):
            raise FusionPicError(-1, ".position <- position and velocity must be [m][3] of one element type")
        ptr = lambda a: None if a is None else a.ptr
        self._check(self._lib.fpic_set_particles_range(self._h, species, int(first), m, ptr(arrs[0]), ptr(arrs[1]), codes.pop()))

    # ---- spatial decomposition (z-slabs; include/fusionpic.h, fpic_domain_*)
    def domainInit(self, rank, world, ghost_planes=2, migrate_every=4, distributed_solve=False):
        # distributed_solve: False / 0 replicated, True / 1 transposed spectrum, "interface" / 2 tridiagonal interface solve along z
        mode = 2 if (distributed_solve == "interface" or (distributed_solve == 2 and distributed_solve is not True)) else (1 if distributed_solve else 0)
        self._check(self._lib.fpic_domain_init(self._h, int(rank), int(world), int(ghost_planes), int(migrate_every), mode))

    def domainSet(self, position, velocity, first_id, species=0):
        p, v = _device_or_host(position), _device_or_host(velocity)
        if p.shape != v.shape or len(p.shape) != 2 or p.shape[1] != 3 or p.code != v.code:
            raise FusionPicError(-1, ".position <- position and velocity must be [n][3] of one element type")
        self._check(self._lib.fpic_domain_set_particles(self._h, species, p.shape[0], p.ptr, v.ptr, int(first_id), p.code))

    def domainGet(self, dtype=None, species=0):
        """{position, velocity, ids} of the particles this rank holds now (no particular order)"""
        code = self.precision if dtype is None else (F32 if np.dtype(dtype) == np.float32 else F64)
        n = ctypes.c_uint64()
        self._check(self._lib.fpic_domain_get_particles(self._h, species, None, None, None, 0, ctypes.byref(n), code))
        m = n.value
        out = {"position": np.empty((m, 3), dtype=_np_dtype(code)), "velocity": np.empty((m, 3), dtype=_np_dtype(code)),
               "ids": np.empty(m, dtype=np.uint32)}
        self._check(self._lib.fpic_domain_get_particles(self._h, species, out["position"].ctypes.data, out["velocity"].ctypes.data,
                                                        out["ids"].ctypes.data, m, ctypes.byref(n), code))
        return out

    def saveCheckpoint(self, path):
        """particles of every species (raw state, caller's order) and the fields; not for a decomposed handle"""
        self._check(self._lib.fpic_save_checkpoint(self._h, os.fsencode(path)))

    def loadCheckpoint(self, path):
        self._check(self._lib.fpic_load_checkpoint(self._h, os.fsencode(path)))

    def domainStats(self):
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.fpic_domain_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return {"migrated": a.value, "lost": b.value}

    def getParticles(self, dtype=None, species=0):
        code = self.precision if dtype is None else (F32 if np.dtype(dtype) == np.float32 else F64)
        n = self.counts[species]
        out = {"position": np.empty((n, 3), dtype=_np_dtype(code)), "velocity": np.empty((n, 3), dtype=_np_dtype(code))}
        self._check(self._lib.fpic_get_particles_of(self._h, species, out["position"].ctypes.data, out["velocity"].ctypes.data, code))
        return out

    def getCells(self, species=0):
        out = np.empty(self.counts[species], dtype=np.int32)
        self._check(self._lib.fpic_get_cells_of(self._h, species, out.ctypes.data))
        return out

    def getRange(self, first, count, stride=1, dtype=None, species=0, cells=False):
        """The caller's particles first, first + stride, ... (`count` of them): the mirror of setRange and a sampled
        read-back (fpic_get_particles_range); cells=True adds their node cells."""
        code = self.precision if dtype is None else (F32 if np.dtype(dtype) == np.float32 else F64)
        out = {"position": np.empty((count, 3), dtype=_np_dtype(code)), "velocity": np.empty((count, 3), dtype=_np_dtype(code))}
        self._check(self._lib.fpic_get_particles_range(self._h, species, int(first), int(count), int(stride),
                                                       out["position"].ctypes.data, out["velocity"].ctypes.data, code))
        if cells:
            out["cells"] = np.empty(count, dtype=np.int32)
            self._check(self._lib.fpic_get_cells_range(self._h, species, int(first), int(count), int(stride), out["cells"].ctypes.data))
        return out

    def readField(self, which, dtype=None):
        """F3_E -> [nodes][4] (Ex, Ey, Ez, phi); F3_RHO / F3_PHI -> [nodes]; F3_RHO_FIXED -> int64 [nodes];
        node index i + nr*(j + ny*k)."""
        if which in (F3_RHO_FIXED, F3_J_FIXED):
            out = np.empty(self.nodes * (3 if which == F3_J_FIXED else 1), dtype=np.int64)
            self._check(self._lib.fpic_read_field3(self._h, which, out.ctypes.data, 0))
            return out.reshape(self.nodes, 3) if which == F3_J_FIXED else out
        code = self.precision if dtype is None else (F32 if np.dtype(dtype) == np.float32 else F64)
        four = which in (F3_E, F3_B_NODES, F3_EDGE_E, F3_FACE_B)
        out = np.empty(self.nodes * (4 if four else 1), dtype=_np_dtype(code))
        self._check(self._lib.fpic_read_field3(self._h, which, out.ctypes.data, code))
        return out.reshape(self.nodes, 4) if four else out


class BoxGroup:
    """All ranks of a z-slab decomposition as handles of this process on one GPU (fpic_group_*): the in-process
    stand-in for the RCCL exchange."""

    def __init__(self, sims):
        self.sims = list(sims)
        self._lib = self.sims[0]._lib
        self._arr = (ctypes.c_void_p * len(self.sims))(*[s._h for s in self.sims])

    def _check(self, rc):
        if rc != 0:
            raise FusionPicError(rc, self._lib.fpic_last_error(self.sims[0]._h).decode())

    def precalc(self):
        self._check(self._lib.fpic_group_precalc(self._arr, len(self.sims)))

    def step(self, ncalls=1):
        self._check(self._lib.fpic_group_step(self._arr, len(self.sims), int(ncalls)))

    def density(self):
        """density() of every member of a full-EM group: the charge grid of the current positions, complete on own planes"""
        self._check(self._lib.fpic_group_density(self._arr, len(self.sims)))


def commUniqueId(library=None):
    """ncclGetUniqueId through the library (rank 0); 128 bytes for commInit on every rank"""
    lib = library or load_library()
    buf = ctypes.create_string_buffer(128)
    rc = lib.fpic_comm_unique_id(buf)
    if rc != 0:
        raise FusionPicError(rc, lib.fpic_last_error(None).decode())
    return buf.raw


def makeCylindricalParticlePusher(spec, **extensions):
    """empic.makeCylindricalParticlePusher(spec) (empic.js:30).  Extension key spec.geometry = 'cart3d'
    selects the self-consistent electrostatic box (no reference counterpart)."""
    if spec.get("geometry", "cyl_rz") == "cart3d":
        return ElectrostaticBoxPusher(spec, **extensions)
    return CylindricalParticlePusher(spec, **extensions)
