/*
 * es3d_oracle_impl.h — body of the CPU restatement of the CART3D electrostatic mode,
 * included once per precision by es3d_oracle.c with REAL / SUF defined.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference has no field solve in its
 * step loop (empic.js:1436-1505 never writes E or B), no Cartesian
 * grid and no CIC shape, so nothing under /root/reference pins this mode.  It is the
 * build's own definition of BASELINE.json configs[2..4], written down once here and
 * followed operation for operation by fusion-sim_amd/csrc/fes_kernels.hpp (both built
 * with -ffp-contract=off).  Where the reference has a counterpart the same convention is
 * kept: positions normalised by the box (empic.js:1199-1244), velocities in units of c,
 * h = q dt / 2m (empic.js:44), step factor dt*c (empic.js:852), 2 sub-steps per step()
 * (empic.js:1436-1469), node index i + nx*(j + ny*k) with i fastest (empic.js:1162).
 *
 * Definitions (T = REAL):
 *   cell / weights of a normalised coordinate u in [0,1) on n nodes:
 *       g = u*n; i = (int)g; f = g - i; if (i >= n) i -= n;        (u*n may round up to n)
 *       w1 = ((int)(f*32768) + 1) >> 1;  w0 = 16384 - w1           (14-bit fixed point, w0+w1 = 2^14 exactly)
 *   deposit: node (i+a, j+b, k+c) += Z * wx[a]*wy[b]*wz[c]  as int64 (exact, order-free);
 *            sum over all nodes = Z * 2^42 * particles, exactly
 *   gather:  E_p = sum over c,b,a (z outermost) of ((fx[a]*fy[b])*fz[c]) * E_node, fx = w * 2^-14 in T; every term is
 *            added with ONE rounding: E_p = fma(w, E_node, E_p)
 *   push:    a = (h/c) E_p;  v- = v + a;  [v' = v- + v- x t;  v+ = v- + v' x s]  (t = hB, s = 2t/(1+t^2));
 *            a cross product's component is fma(p, q, -(r*s));  v_new = v+ + a;
 *            u_new = wrap(fma(dt c / L, v_new, u));  wrap(u) = u - floor(u), 1 -> 0
 *   Fused multiply-adds are part of the DEFINITION (round 3): they are written out as fma() here and in the kernels,
 *   nothing is left to a compiler's contraction (-ffp-contract=off on both sides); they halve the arithmetic
 *   instructions of the gather, which is what bounds the push kernels next to their HBM streams.
 *   solve:   phi_hat = rho_hat / (eps0 K^2),  K^2 = sum_axis (2/d sin(pi l/n))^2,  mean mode = 0
 *            E_x[i] = (phi[i-1] - phi[i+1]) * (1/(2 dx))   (periodic), node record (Ex, Ey, Ez, phi)
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

static inline void FN(es3d_axis)(REAL u, int n, int* i0, int* w1)
{
    const REAL g = u * (REAL)n;
    int i = (int)g;
    const REAL f = g - (REAL)i;
    if (i >= n) i -= n;
    *i0 = i;
    *w1 = ((int)(f * (REAL)32768) + 1) >> 1;
}

/* u - floor(u), with the one value that can round to 1 folded to 0 */
static inline REAL FN(es3d_wrap)(REAL u)
{
    REAL r = u - FN_FLOOR(u);
    if (!(r < (REAL)1)) r = (REAL)0;
    return r;
}

/* out.set({position}) for the box: u = (T)(x * (1/L)) (one rounding, empic.js:1199-1244), wrapped */
void FN(es3d_normalise)(const double* pos_aos, size_t n, double lx, double ly, double lz, REAL* x, REAL* y, REAL* z)
{
    const double fx = 1 / lx, fy = 1 / ly, fz = 1 / lz;
    for (size_t p = 0; p < n; ++p) {
        x[p] = FN(es3d_wrap)((REAL)(pos_aos[3 * p] * fx));
        y[p] = FN(es3d_wrap)((REAL)(pos_aos[3 * p + 1] * fy));
        z[p] = FN(es3d_wrap)((REAL)(pos_aos[3 * p + 2] * fz));
    }
}

/* CIC deposit of charge number Z per particle into the int64 node grid (accumulates) */
void FN(es3d_deposit)(const REAL* x, const REAL* y, const REAL* z, size_t n, int nx, int ny, int nz, int Z, int64_t* rho)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n; ++p) {
        int i, j, k, wx[2], wy[2], wz[2];
        FN(es3d_axis)(x[p], nx, &i, &wx[1]); wx[0] = 16384 - wx[1];
        FN(es3d_axis)(y[p], ny, &j, &wy[1]); wy[0] = 16384 - wy[1];
        FN(es3d_axis)(z[p], nz, &k, &wz[1]); wz[0] = 16384 - wz[1];
        for (int c = 0; c < 2; ++c)
            for (int b = 0; b < 2; ++b)
                for (int a = 0; a < 2; ++a) {
                    const int ii = (i + a == nx) ? 0 : i + a, jj = (j + b == ny) ? 0 : j + b, kk = (k + c == nz) ? 0 : k + c;
                    const int64_t w = (int64_t)wx[a] * wy[b] * wz[c] * Z;
                    int64_t* t = rho + ((size_t)ii + (size_t)nx * ((size_t)jj + (size_t)ny * kk));
#ifdef _OPENMP
#pragma omp atomic
#endif
                    *t += w;
                }
    }
}

/* NGP cell index i + nx*(j + ny*k) of every particle (integer parity check) */
void FN(es3d_cells)(const REAL* x, const REAL* y, const REAL* z, size_t n, int nx, int ny, int nz, int32_t* cells)
{
    for (size_t p = 0; p < n; ++p) {
        int i, j, k, w;
        FN(es3d_axis)(x[p], nx, &i, &w);
        FN(es3d_axis)(y[p], ny, &j, &w);
        FN(es3d_axis)(z[p], nz, &k, &w);
        cells[p] = (int32_t)(i + nx * (j + ny * k));
    }
}

/* rho[node] = (T)((double)fixed * scale), scale = q0 W / (2^42 dV) */
void FN(es3d_rho_real)(const int64_t* fixed, size_t nodes, double scale, REAL* rho)
{
    for (size_t c = 0; c < nodes; ++c) rho[c] = (REAL)((double)fixed[c] * scale);
}

/* Poisson solve in double (the "truth" the GPU's T-precision FFT is held to within a tolerance), phi cast to T */
void FN(es3d_poisson)(const REAL* rho, int nx, int ny, int nz, double lx, double ly, double lz, REAL* phi)
{
    const size_t N = (size_t)nx * ny * nz;
    double* re = (double*)malloc(sizeof(double) * N);
    double* im = (double*)calloc(N, sizeof(double));
    for (size_t c = 0; c < N; ++c) re[c] = (double)rho[c];
    es3d_fft3(re, im, nx, ny, nz, -1);
    double* k2x = es3d_k2_table(nx, lx / nx);
    double* k2y = es3d_k2_table(ny, ly / ny);
    double* k2z = es3d_k2_table(nz, lz / nz);
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t c = (size_t)i + (size_t)nx * ((size_t)j + (size_t)ny * k);
                const double K2 = (k2x[i] + k2y[j]) + k2z[k];
                const double g = (i | j | k) ? 1.0 / (ES3D_EPS0 * K2 * (double)N) : 0.0;
                re[c] *= g; im[c] *= g;
            }
    es3d_fft3(re, im, nx, ny, nz, +1);
    for (size_t c = 0; c < N; ++c) phi[c] = (REAL)re[c];
    free(re); free(im); free(k2x); free(k2y); free(k2z);
}

/* node records (Ex, Ey, Ez, phi) from phi by central differences, periodic */
void FN(es3d_gradient)(const REAL* phi, int nx, int ny, int nz, double lx, double ly, double lz, REAL* E4)
{
    const REAL hx = (REAL)(1.0 / (2.0 * (lx / nx))), hy = (REAL)(1.0 / (2.0 * (ly / ny))), hz = (REAL)(1.0 / (2.0 * (lz / nz)));
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
                const size_t c = (size_t)i + sy * j + sz * k;
                const int im = i ? i - 1 : nx - 1, ip = (i + 1 == nx) ? 0 : i + 1;
                const int jm = j ? j - 1 : ny - 1, jp = (j + 1 == ny) ? 0 : j + 1;
                const int km = k ? k - 1 : nz - 1, kp = (k + 1 == nz) ? 0 : k + 1;
                E4[4 * c] = (phi[(size_t)im + sy * j + sz * k] - phi[(size_t)ip + sy * j + sz * k]) * hx;
                E4[4 * c + 1] = (phi[(size_t)i + sy * jm + sz * k] - phi[(size_t)i + sy * jp + sz * k]) * hy;
                E4[4 * c + 2] = (phi[(size_t)i + sy * j + sz * km] - phi[(size_t)i + sy * j + sz * kp]) * hz;
                E4[4 * c + 3] = phi[c];
            }
}

/* one leap-frog sub-step of every particle: CIC gather of E, Boris, drift, periodic wrap.
 * par = { h/c, tx, ty, tz, sx, sy, sz, dt c / lx, dt c / ly, dt c / lz } already in T */
void FN(es3d_push)(REAL* x, REAL* y, REAL* z, REAL* vx, REAL* vy, REAL* vz, size_t n, const REAL* E4, int nx, int ny, int nz,
                   const REAL* par, int has_b)
{
    const REAL hc = par[0], tx = par[1], ty = par[2], tz = par[3], sx = par[4], sy = par[5], sz = par[6];
    const REAL dx = par[7], dy = par[8], dz = par[9];
    const REAL q14 = (REAL)(1.0 / 16384.0);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n; ++p) {
        int i, j, k, w1;
        REAL fx[2], fy[2], fz[2];
        FN(es3d_axis)(x[p], nx, &i, &w1); fx[1] = (REAL)w1 * q14; fx[0] = (REAL)(16384 - w1) * q14;
        FN(es3d_axis)(y[p], ny, &j, &w1); fy[1] = (REAL)w1 * q14; fy[0] = (REAL)(16384 - w1) * q14;
        FN(es3d_axis)(z[p], nz, &k, &w1); fz[1] = (REAL)w1 * q14; fz[0] = (REAL)(16384 - w1) * q14;
        REAL Ex = 0, Ey = 0, Ez = 0;
        for (int c = 0; c < 2; ++c)
            for (int b = 0; b < 2; ++b)
                for (int a = 0; a < 2; ++a) {
                    const int ii = (i + a == nx) ? 0 : i + a, jj = (j + b == ny) ? 0 : j + b, kk = (k + c == nz) ? 0 : k + c;
                    const REAL* e = E4 + 4 * ((size_t)ii + (size_t)nx * ((size_t)jj + (size_t)ny * kk));
                    const REAL w = (fx[a] * fy[b]) * fz[c];
                    Ex = FN_FMA(w, e[0], Ex);
                    Ey = FN_FMA(w, e[1], Ey);
                    Ez = FN_FMA(w, e[2], Ez);
                }
        const REAL ax = hc * Ex, ay = hc * Ey, az = hc * Ez;
        REAL ux = vx[p] + ax, uy = vy[p] + ay, uz = vz[p] + az;
        if (has_b) {
            const REAL px = ux + FN_FMA(uy, tz, -(uz * ty));
            const REAL py = uy + FN_FMA(uz, tx, -(ux * tz));
            const REAL pz = uz + FN_FMA(ux, ty, -(uy * tx));
            const REAL qx = ux + FN_FMA(py, sz, -(pz * sy));
            const REAL qy = uy + FN_FMA(pz, sx, -(px * sz));
            const REAL qz = uz + FN_FMA(px, sy, -(py * sx));
            ux = qx; uy = qy; uz = qz;
        }
        const REAL nvx = ux + ax, nvy = uy + ay, nvz = uz + az;
        vx[p] = nvx; vy[p] = nvy; vz[p] = nvz;
        x[p] = FN(es3d_wrap)(FN_FMA(dx, nvx, x[p]));
        y[p] = FN(es3d_wrap)(FN_FMA(dy, nvy, y[p]));
        z[p] = FN(es3d_wrap)(FN_FMA(dz, nvz, z[p]));
    }
}


/* ================================================================ full EM (Yee FDTD), BASELINE configs[4]
 *
 * PARITY UNPINNED (no reference counterpart).  Fields live on the Yee lattice, stored 4 T per node:
 *   Ey[node] = (Ex(i+1/2,j,k), Ey(i,j+1/2,k), Ez(i,j,k+1/2), 0),  By[node] = (Bx(i,j+1/2,k+1/2), By(i+1/2,j,k+1/2), Bz(i+1/2,j+1/2,k), 0).
 * One sub-step (E^n, B^n, x^n, v^(n-1/2) -> n+1):
 *   nodes:  E4n, B4n = the lattice values averaged to the nodes (2 resp. 4 neighbours, periodic)
 *   push:   CIC gather of E4n and B4n, Boris with the particle's own t = hB, s = 2t/(1+t^2), drift, wrap
 *   J:      charge-conserving zigzag deposit in integers (es3d_current)
 *   B^(n+1/2) = B^n - dt/2 curl E^n;  E^(n+1) = E^n + dt (c^2 curl B^(n+1/2) - J/eps0);  B^(n+1) = B^(n+1/2) - dt/2 curl E^(n+1)
 */

/* node-centred copies of the staggered fields */
void FN(em_nodes)(const REAL* Ey, const REAL* By, int nx, int ny, int nz, REAL* E4n, REAL* B4n)
{
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t im = i ? i - 1 : nx - 1, jm = j ? j - 1 : ny - 1, km = k ? k - 1 : nz - 1;
                const size_t c = i + sy * j + sz * k;
#define AT(A, ii, jj, kk, comp) A[4 * ((size_t)(ii) + sy * (size_t)(jj) + sz * (size_t)(kk)) + (comp)]
                E4n[4 * c] = (REAL)0.5 * (AT(Ey, im, j, k, 0) + AT(Ey, i, j, k, 0));
                E4n[4 * c + 1] = (REAL)0.5 * (AT(Ey, i, jm, k, 1) + AT(Ey, i, j, k, 1));
                E4n[4 * c + 2] = (REAL)0.5 * (AT(Ey, i, j, km, 2) + AT(Ey, i, j, k, 2));
                E4n[4 * c + 3] = (REAL)0;
                B4n[4 * c] = (REAL)0.25 * (((AT(By, i, jm, km, 0) + AT(By, i, j, km, 0)) + AT(By, i, jm, k, 0)) + AT(By, i, j, k, 0));
                B4n[4 * c + 1] = (REAL)0.25 * (((AT(By, im, j, km, 1) + AT(By, i, j, km, 1)) + AT(By, im, j, k, 1)) + AT(By, i, j, k, 1));
                B4n[4 * c + 2] = (REAL)0.25 * (((AT(By, im, jm, k, 2) + AT(By, i, jm, k, 2)) + AT(By, im, j, k, 2)) + AT(By, i, j, k, 2));
                B4n[4 * c + 3] = (REAL)0;
            }
}

/* one sub-step of every particle in the node-centred fields; the old positions are kept in ox, oy, oz for the
 * current deposit.  par = { h (not /c), c, dt c / lx, dt c / ly, dt c / lz }: E in V/m, B in T, v in units of c */
void FN(em_push)(REAL* x, REAL* y, REAL* z, REAL* vx, REAL* vy, REAL* vz, REAL* ox, REAL* oy, REAL* oz, size_t n, const REAL* E4n,
                 const REAL* B4n, int nx, int ny, int nz, const REAL* par)
{
    const REAL h = par[0], hc = par[0] / par[1], dx = par[2], dy = par[3], dz = par[4];
    const REAL q14 = (REAL)(1.0 / 16384.0);
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n; ++p) {
        int i, j, k, w1;
        REAL fx[2], fy[2], fz[2];
        FN(es3d_axis)(x[p], nx, &i, &w1); fx[1] = (REAL)w1 * q14; fx[0] = (REAL)(16384 - w1) * q14;
        FN(es3d_axis)(y[p], ny, &j, &w1); fy[1] = (REAL)w1 * q14; fy[0] = (REAL)(16384 - w1) * q14;
        FN(es3d_axis)(z[p], nz, &k, &w1); fz[1] = (REAL)w1 * q14; fz[0] = (REAL)(16384 - w1) * q14;
        REAL E[3] = { 0, 0, 0 }, B[3] = { 0, 0, 0 };
        for (int c = 0; c < 2; ++c)
            for (int b = 0; b < 2; ++b)
                for (int a = 0; a < 2; ++a) {
                    const int ii = (i + a == nx) ? 0 : i + a, jj = (j + b == ny) ? 0 : j + b, kk = (k + c == nz) ? 0 : k + c;
                    const size_t node = (size_t)ii + (size_t)nx * ((size_t)jj + (size_t)ny * kk);
                    const REAL w = (fx[a] * fy[b]) * fz[c];
                    for (int m = 0; m < 3; ++m) {
                        E[m] = FN_FMA(w, E4n[4 * node + m], E[m]);
                        B[m] = FN_FMA(w, B4n[4 * node + m], B[m]);
                    }
                }
        const REAL ax = hc * E[0], ay = hc * E[1], az = hc * E[2];
        const REAL tx = h * B[0], ty = h * B[1], tz = h * B[2];
        const REAL f = (REAL)2 / ((REAL)1 + FN_FMA(tz, tz, FN_FMA(ty, ty, tx * tx)));
        const REAL sx = f * tx, sy = f * ty, sz = f * tz;
        const REAL ux = vx[p] + ax, uy = vy[p] + ay, uz = vz[p] + az;
        const REAL px = ux + FN_FMA(uy, tz, -(uz * ty));
        const REAL py = uy + FN_FMA(uz, tx, -(ux * tz));
        const REAL pz = uz + FN_FMA(ux, ty, -(uy * tx));
        const REAL qx = ux + FN_FMA(py, sz, -(pz * sy));
        const REAL qy = uy + FN_FMA(pz, sx, -(px * sz));
        const REAL qz = uz + FN_FMA(px, sy, -(py * sx));
        const REAL nvx = qx + ax, nvy = qy + ay, nvz = qz + az;
        vx[p] = nvx; vy[p] = nvy; vz[p] = nvz;
        ox[p] = x[p]; oy[p] = y[p]; oz[p] = z[p];
        x[p] = FN(es3d_wrap)(FN_FMA(dx, nvx, x[p]));
        y[p] = FN(es3d_wrap)(FN_FMA(dy, nvy, y[p]));
        z[p] = FN(es3d_wrap)(FN_FMA(dz, nvz, z[p]));
    }
}

/* doubled fixed-point lattice coordinate of a normalised coordinate: 2 * (cell * 2^14 + w1) in [0, n * 2^15) */
static inline int64_t FN(em_coord)(REAL u, int n)
{
    int i, w1;
    FN(es3d_axis)(u, n, &i, &w1);
    return 2 * ((int64_t)i * 16384 + w1);
}

/* charge-conserving current of the moves (ox,oy,oz) -> (x,y,z), accumulated into Jfix (3 int64 per node, see es3d_current) */
void FN(em_current)(const REAL* ox, const REAL* oy, const REAL* oz, const REAL* x, const REAL* y, const REAL* z, size_t n, int nx, int ny,
                    int nz, int Z, int64_t* Jfix)
{
    for (size_t p = 0; p < n; ++p) {
        const int64_t a[3] = { FN(em_coord)(ox[p], nx), FN(em_coord)(oy[p], ny), FN(em_coord)(oz[p], nz) };
        const int64_t b[3] = { FN(em_coord)(x[p], nx), FN(em_coord)(y[p], ny), FN(em_coord)(z[p], nz) };
        es3d_current(a, b, nx, ny, nz, Z, Jfix);
    }
}

/* J (A/m^2, 4 T per node: Jx on the x-edge, ...) from the integer grid: T((double)fixed * scale[comp]) */
void FN(em_j_real)(const int64_t* Jfix, size_t nodes, const double* scale3, REAL* J4)
{
    for (size_t c = 0; c < nodes; ++c) {
        for (int m = 0; m < 3; ++m) J4[4 * c + m] = (REAL)((double)Jfix[3 * c + m] * scale3[m]);
        J4[4 * c + 3] = (REAL)0;
    }
}

/* B -= coef * curl E on the Yee lattice (coef = dt/2 over the spacings: cb = { dt/2dx, dt/2dy, dt/2dz }) */
void FN(em_update_b)(REAL* By, const REAL* Ey, int nx, int ny, int nz, const REAL* cb)
{
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1, kp = (k + 1 == nz) ? 0 : k + 1;
                const size_t c = i + sy * j + sz * k;
                /* (curl E)_x at (i, j+1/2, k+1/2) = dEz/dy - dEy/dz */
                const REAL cx = (AT(Ey, i, jp, k, 2) - AT(Ey, i, j, k, 2)) * cb[1] - (AT(Ey, i, j, kp, 1) - AT(Ey, i, j, k, 1)) * cb[2];
                const REAL cy = (AT(Ey, i, j, kp, 0) - AT(Ey, i, j, k, 0)) * cb[2] - (AT(Ey, ip, j, k, 2) - AT(Ey, i, j, k, 2)) * cb[0];
                const REAL cz = (AT(Ey, ip, j, k, 1) - AT(Ey, i, j, k, 1)) * cb[0] - (AT(Ey, i, jp, k, 0) - AT(Ey, i, j, k, 0)) * cb[1];
                By[4 * c] = By[4 * c] - cx;
                By[4 * c + 1] = By[4 * c + 1] - cy;
                By[4 * c + 2] = By[4 * c + 2] - cz;
            }
}

/* E += ce * curl B - je * J  (ce = { c^2 dt/dx, c^2 dt/dy, c^2 dt/dz }, je = dt/eps0) */
void FN(em_update_e)(REAL* Ey, const REAL* By, const REAL* J4, int nx, int ny, int nz, const REAL* ce, REAL je)
{
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t im = i ? i - 1 : nx - 1, jm = j ? j - 1 : ny - 1, km = k ? k - 1 : nz - 1;
                const size_t c = i + sy * j + sz * k;
                /* (curl B)_x at (i+1/2, j, k) = dBz/dy - dBy/dz */
                const REAL cx = (AT(By, i, j, k, 2) - AT(By, i, jm, k, 2)) * ce[1] - (AT(By, i, j, k, 1) - AT(By, i, j, km, 1)) * ce[2];
                const REAL cy = (AT(By, i, j, k, 0) - AT(By, i, j, km, 0)) * ce[2] - (AT(By, i, j, k, 2) - AT(By, im, j, k, 2)) * ce[0];
                const REAL cz = (AT(By, i, j, k, 1) - AT(By, im, j, k, 1)) * ce[0] - (AT(By, i, j, k, 0) - AT(By, i, jm, k, 0)) * ce[1];
                Ey[4 * c] = (Ey[4 * c] + cx) - je * J4[4 * c];
                Ey[4 * c + 1] = (Ey[4 * c + 1] + cy) - je * J4[4 * c + 1];
                Ey[4 * c + 2] = (Ey[4 * c + 2] + cz) - je * J4[4 * c + 2];
            }
}
#undef AT

/* E on the Yee edges from phi: Ex(i+1/2) = (phi[i] - phi[i+1]) / dx: with the 3-point Laplacian of es3d_poisson this
 * satisfies the lattice's Gauss law exactly (div E = rho / eps0), the state a charge-conserving run keeps */
void FN(em_edge_gradient)(const REAL* phi, int nx, int ny, int nz, double lx, double ly, double lz, REAL* Ey)
{
    const REAL hx = (REAL)(1.0 / (lx / nx)), hy = (REAL)(1.0 / (ly / ny)), hz = (REAL)(1.0 / (lz / nz));
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1, kp = (k + 1 == nz) ? 0 : k + 1;
                const size_t c = i + sy * j + sz * k;
                Ey[4 * c] = (phi[c] - phi[ip + sy * j + sz * k]) * hx;
                Ey[4 * c + 1] = (phi[c] - phi[i + sy * jp + sz * k]) * hy;
                Ey[4 * c + 2] = (phi[c] - phi[i + sy * j + sz * kp]) * hz;
                Ey[4 * c + 3] = (REAL)0;
            }
}

#undef FN
#undef CAT
#undef CAT_
