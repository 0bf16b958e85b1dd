"""ctypes front-end of the CART3D electrostatic oracle (oracle/es3d_oracle.c).

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (the reference has no self-consistent mode,
see es3d_oracle_impl.h).  `OracleES3D` mirrors the object the product factory returns
for spec.geometry == 'cart3d' (fusionpic.ElectrostaticBoxPusher): same spec keys, same
method names, same state.
"""
import ctypes

import numpy as np

import pic_oracle

E_CHARGE = 1.602e-19          # the demo's proton charge (fusionsim.js:81)
SPEED_OF_LIGHT = 2.998e8      # empic.js:27
EPS0 = 8.8541878128e-12
FIXED_ONE = 1 << 42           # one particle of charge number 1 deposits exactly this much in total


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Species:
    def __init__(self, mass, charge, n, dtype):
        self.mass, self.charge, self.n = float(mass), float(charge), int(n)
        self.x, self.y, self.z, self.vx, self.vy, self.vz = (np.zeros(self.n, dtype=dtype) for _ in range(6))


class OracleES3D:
    """Periodic box lx x ly x lz = (radius, length_y, height), nodes nx x ny x nz = (nr, ny, nz)."""

    def __init__(self, spec, dtype=np.float32, threads=0):
        self.lib = pic_oracle.lib_omp() if threads else pic_oracle.lib()
        if threads:
            self.lib.es3d_set_threads(int(threads))
        self.dtype = np.dtype(dtype)
        self.suf = "_f32" if self.dtype == np.float32 else "_f64"
        self.spec = dict(spec)
        self.nx, self.ny, self.nz = int(spec["nr"]), int(spec["ny"]), int(spec["nz"])
        self.lx, self.ly, self.lz = float(spec["radius"]), float(spec["length_y"]), float(spec["height"])
        self.dt = float(spec["dt"])
        self.W = float(spec.get("macro_weight", 1.0))
        self.solver = spec.get("solver", "poisson_fft")
        n0 = int(spec.get("count") or int(spec["nparticles"]) ** 2)
        self.species = [Species(spec["particle_mass"], spec["particle_charge"], n0, self.dtype)]
        self.q0 = float(spec["particle_charge"])
        self.nodes = self.nx * self.ny * self.nz
        self.rho_fixed = np.zeros(self.nodes, dtype=np.int64)
        self.rho = np.zeros(self.nodes, dtype=self.dtype)
        self.phi = np.zeros(self.nodes, dtype=self.dtype)
        self.E4 = np.zeros(4 * self.nodes, dtype=self.dtype)
        self.B0 = np.zeros(3)
        self.t = 0
        # full EM (solver 'yee'): staggered fields 4 T per node, node-centred copies, integer current grid
        if self.solver == "yee":
            self.Ey = np.zeros(4 * self.nodes, dtype=self.dtype)
            self.By = np.zeros(4 * self.nodes, dtype=self.dtype)
            self.B4n = np.zeros(4 * self.nodes, dtype=self.dtype)
            self.J4 = np.zeros(4 * self.nodes, dtype=self.dtype)
            self.J_fixed = np.zeros(3 * self.nodes, dtype=np.int64)

    def _fn(self, name):
        return getattr(self.lib, name + self.suf)

    # ---- host surface
    def add_species(self, mass, charge, count):
        z = charge / self.q0
        if abs(z - round(z)) > 1e-6 or round(z) == 0 or abs(round(z)) > 255:
            raise ValueError(".charge <- must be a non-zero integer multiple (|Z| <= 255) of species 0's charge")
        self.species.append(Species(mass, charge, count, self.dtype))
        return len(self.species) - 1

    def charge_number(self, s):
        return int(round(self.species[s].charge / self.q0))

    def set(self, position=None, velocity=None, E=None, species=0):
        sp = self.species[species]
        if position is not None:
            p = np.ascontiguousarray(position, dtype=np.float64)
            assert p.shape == (sp.n, 3)
            self._fn("es3d_normalise")(_ptr(p), ctypes.c_size_t(sp.n), ctypes.c_double(self.lx), ctypes.c_double(self.ly),
                                       ctypes.c_double(self.lz), _ptr(sp.x), _ptr(sp.y), _ptr(sp.z))
        if velocity is not None:
            v = np.asarray(velocity, dtype=np.float64)
            assert v.shape == (sp.n, 3)
            sp.vx[:], sp.vy[:], sp.vz[:] = (v[:, k].astype(self.dtype) for k in range(3))
        if E is not None:  # [nx][ny][nz][3] V/m, static field (solver 'none') or a field injected for a parity test
            e = np.asarray(E, dtype=np.float64)
            assert e.shape == (self.nx, self.ny, self.nz, 3)
            e4 = self.E4.reshape(self.nz, self.ny, self.nx, 4)
            e4[..., :3] = e.transpose(2, 1, 0, 3).astype(self.dtype)

    def add_b(self, bx, by, bz):
        self.B0 += np.array([bx, by, bz], dtype=np.float64)

    def add_bz(self, bz):
        self.add_b(0.0, 0.0, bz)

    def push_params(self, s):
        """par[10] of es3d_push, derived in double and rounded once into T (the product derives the same)."""
        sp = self.species[s]
        h = sp.charge * self.dt / (2 * sp.mass)                   # empic.js:44
        t = h * self.B0
        sv = 2 * t / (1 + float(t @ t))
        step = self.dt * SPEED_OF_LIGHT                           # empic.js:852
        par = np.array([h / SPEED_OF_LIGHT, t[0], t[1], t[2], sv[0], sv[1], sv[2], step / self.lx, step / self.ly, step / self.lz])
        return par.astype(self.dtype), bool(np.any(self.B0 != 0))

    def rho_scale(self):
        dv = (self.lx / self.nx) * (self.ly / self.ny) * (self.lz / self.nz)
        return self.q0 * self.W / (FIXED_ONE * dv)

    def deposit(self):
        self.rho_fixed[:] = 0
        for s, sp in enumerate(self.species):
            self._fn("es3d_deposit")(_ptr(sp.x), _ptr(sp.y), _ptr(sp.z), ctypes.c_size_t(sp.n), self.nx, self.ny, self.nz,
                                     self.charge_number(s), _ptr(self.rho_fixed))

    def solve(self):
        self._fn("es3d_rho_real")(_ptr(self.rho_fixed), ctypes.c_size_t(self.nodes), ctypes.c_double(self.rho_scale()), _ptr(self.rho))
        if self.solver != "poisson_fft":
            return
        self._fn("es3d_poisson")(_ptr(self.rho), self.nx, self.ny, self.nz, ctypes.c_double(self.lx), ctypes.c_double(self.ly),
                                 ctypes.c_double(self.lz), _ptr(self.phi))
        self._fn("es3d_gradient")(_ptr(self.phi), self.nx, self.ny, self.nz, ctypes.c_double(self.lx), ctypes.c_double(self.ly),
                                  ctypes.c_double(self.lz), _ptr(self.E4))

    def precalc(self):
        """fields <- particles: deposit and solve (the stage the reference's precalc() stands for, empic.js:1413)."""
        self.deposit()
        if self.solver == "yee":
            # initial condition of the EM run: the electrostatic field of the charge on the Yee edges (the lattice's
            # Gauss law holds exactly and the charge-conserving current keeps it), B = the uniform external field
            d = ctypes.c_double
            self._fn("es3d_rho_real")(_ptr(self.rho_fixed), ctypes.c_size_t(self.nodes), d(self.rho_scale()), _ptr(self.rho))
            self._fn("es3d_poisson")(_ptr(self.rho), self.nx, self.ny, self.nz, d(self.lx), d(self.ly), d(self.lz), _ptr(self.phi))
            self._fn("em_edge_gradient")(_ptr(self.phi), self.nx, self.ny, self.nz, d(self.lx), d(self.ly), d(self.lz), _ptr(self.Ey))
            b = self.By.reshape(-1, 4)
            b[:, :3] = self.B0.astype(self.dtype)
            self._fn("em_nodes")(_ptr(self.Ey), _ptr(self.By), self.nx, self.ny, self.nz, _ptr(self.E4), _ptr(self.B4n))
            return
        self.solve()

    # ---- full EM
    def em_constants(self):
        dt = self.dt
        d = (self.lx / self.nx, self.ly / self.ny, self.lz / self.nz)
        cb = np.array([dt / (2 * d[0]), dt / (2 * d[1]), dt / (2 * d[2])]).astype(self.dtype)
        c2 = SPEED_OF_LIGHT ** 2
        ce = np.array([c2 * dt / d[0], c2 * dt / d[1], c2 * dt / d[2]]).astype(self.dtype)
        je = self.dtype.type(dt / EPS0)
        # J = q0 W / (96 * 2^42 dt) * flux / (area of the dual face)
        base = self.q0 * self.W / (96.0 * FIXED_ONE * dt)
        jscale = np.array([base / (d[1] * d[2]), base / (d[0] * d[2]), base / (d[0] * d[1])])
        return cb, ce, je, jscale

    def em_substep(self):
        cb, ce, je, jscale = self.em_constants()
        self._fn("em_nodes")(_ptr(self.Ey), _ptr(self.By), self.nx, self.ny, self.nz, _ptr(self.E4), _ptr(self.B4n))
        self.J_fixed[:] = 0
        for s, sp in enumerate(self.species):
            h = sp.charge * self.dt / (2 * sp.mass)
            step = self.dt * SPEED_OF_LIGHT
            par = np.array([h, SPEED_OF_LIGHT, step / self.lx, step / self.ly, step / self.lz]).astype(self.dtype)
            ox, oy, oz = (np.empty_like(sp.x) for _ in range(3))
            self._fn("em_push")(_ptr(sp.x), _ptr(sp.y), _ptr(sp.z), _ptr(sp.vx), _ptr(sp.vy), _ptr(sp.vz), _ptr(ox), _ptr(oy), _ptr(oz),
                                ctypes.c_size_t(sp.n), _ptr(self.E4), _ptr(self.B4n), self.nx, self.ny, self.nz, _ptr(par))
            self._fn("em_current")(_ptr(ox), _ptr(oy), _ptr(oz), _ptr(sp.x), _ptr(sp.y), _ptr(sp.z), ctypes.c_size_t(sp.n), self.nx, self.ny,
                                   self.nz, self.charge_number(s), _ptr(self.J_fixed))
        self._fn("em_j_real")(_ptr(self.J_fixed), ctypes.c_size_t(self.nodes), _ptr(jscale), _ptr(self.J4))
        self._fn("em_update_b")(_ptr(self.By), _ptr(self.Ey), self.nx, self.ny, self.nz, _ptr(cb))
        self._fn("em_update_e")(_ptr(self.Ey), _ptr(self.By), _ptr(self.J4), self.nx, self.ny, self.nz, _ptr(ce), ctypes.c_float(je) if self.dtype == np.float32 else ctypes.c_double(je))
        self._fn("em_update_b")(_ptr(self.By), _ptr(self.Ey), self.nx, self.ny, self.nz, _ptr(cb))
        self.t += 1

    def set_lattice(self, E=None, B=None):
        """the Yee lattice's own arrays, value[i][j][k][3]: E on the edges (Ex at (i+1/2,j,k), ...), B on the faces"""
        for src, dst in ((E, self.Ey), (B, self.By)):
            if src is not None:
                a = np.asarray(src, dtype=np.float64)
                assert a.shape == (self.nx, self.ny, self.nz, 3)
                dst.reshape(self.nz, self.ny, self.nx, 4)[..., :3] = a.transpose(2, 1, 0, 3).astype(self.dtype)
        self._fn("em_nodes")(_ptr(self.Ey), _ptr(self.By), self.nx, self.ny, self.nz, _ptr(self.E4), _ptr(self.B4n))

    def gauss_residual(self):
        """div E on the lattice minus (rho - mean rho) / eps0, per node (rho of the CURRENT positions is deposited here;
        the mean is the neutralising background: the Poisson solve drops the mean mode)"""
        self.deposit()
        e = self.Ey.reshape(self.nz, self.ny, self.nx, 4).astype(np.float64)
        d = (self.lx / self.nx, self.ly / self.ny, self.lz / self.nz)
        div = ((e[..., 0] - np.roll(e[..., 0], 1, axis=2)) / d[0] + (e[..., 1] - np.roll(e[..., 1], 1, axis=1)) / d[1]
               + (e[..., 2] - np.roll(e[..., 2], 1, axis=0)) / d[2])
        rho = self.rho_fixed.reshape(self.nz, self.ny, self.nx).astype(np.float64) * self.rho_scale()
        return div - (rho - rho.mean()) / EPS0, np.abs(rho).max() / EPS0

    def em_field_energy(self):
        dv = (self.lx / self.nx) * (self.ly / self.ny) * (self.lz / self.nz)
        e = self.Ey.reshape(-1, 4)[:, :3].astype(np.float64)
        b = self.By.reshape(-1, 4)[:, :3].astype(np.float64)
        mu0 = 1.0 / (EPS0 * SPEED_OF_LIGHT ** 2)
        return 0.5 * EPS0 * float((e ** 2).sum()) * dv + 0.5 / mu0 * float((b ** 2).sum()) * dv

    def lattice_fields(self):
        """(E, B) on the Yee lattice as [nx][ny][nz][3]"""
        e = self.Ey.reshape(self.nz, self.ny, self.nx, 4)[..., :3].transpose(2, 1, 0, 3).copy()
        b = self.By.reshape(self.nz, self.ny, self.nx, 4)[..., :3].transpose(2, 1, 0, 3).copy()
        return e, b

    def push(self):
        for s, sp in enumerate(self.species):
            par, has_b = self.push_params(s)
            self._fn("es3d_push")(_ptr(sp.x), _ptr(sp.y), _ptr(sp.z), _ptr(sp.vx), _ptr(sp.vy), _ptr(sp.vz), ctypes.c_size_t(sp.n),
                                  _ptr(self.E4), self.nx, self.ny, self.nz, _ptr(par), int(has_b))

    def substep(self):
        if self.solver == "yee":
            self.em_substep()
            return
        self.push()
        self.deposit()
        self.solve()
        self.t += 1

    def step(self, ncalls=1):
        """one call = two leap-frog sub-steps (empic.js:1436-1469), each: push, deposit, solve"""
        for _ in range(2 * int(ncalls)):
            self.substep()

    # ---- read-back
    def positions(self, s=0):
        sp = self.species[s]
        return np.stack([sp.x, sp.y, sp.z], axis=1)

    def velocities(self, s=0):
        sp = self.species[s]
        return np.stack([sp.vx, sp.vy, sp.vz], axis=1)

    def cells(self, s=0):
        sp = self.species[s]
        out = np.empty(sp.n, dtype=np.int32)
        self._fn("es3d_cells")(_ptr(sp.x), _ptr(sp.y), _ptr(sp.z), ctypes.c_size_t(sp.n), self.nx, self.ny, self.nz, _ptr(out))
        return out

    def field(self):
        """E as [nx][ny][nz][3] and phi as [nx][ny][nz]"""
        e4 = self.E4.reshape(self.nz, self.ny, self.nx, 4)
        return e4[..., :3].transpose(2, 1, 0, 3).copy(), e4[..., 3].transpose(2, 1, 0).copy()

    def field_energy(self):
        e4 = self.E4.reshape(-1, 4).astype(np.float64)
        dv = (self.lx / self.nx) * (self.ly / self.ny) * (self.lz / self.nz)
        return 0.5 * EPS0 * float((e4[:, :3] ** 2).sum()) * dv

    def kinetic_energy(self):
        tot = 0.0
        for sp in self.species:
            v2 = sp.vx.astype(np.float64) ** 2 + sp.vy.astype(np.float64) ** 2 + sp.vz.astype(np.float64) ** 2
            tot += 0.5 * sp.mass * self.W * SPEED_OF_LIGHT ** 2 * float(v2.sum())
        return tot
