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
 *   gather:  E_p = sum over c,b,a (z outermost) of ((fx[a]*fy[b])*fz[c]) * E_node, fx = w * 2^-14 in T
 *   push:    a = (h/c) E_p;  v- = v + a;  [v' = v- + v- x t;  v+ = v- + v' x s]  (t = hB, s = 2t/(1+t^2));
 *            v_new = v+ + a;  u_new = wrap(u + (dt c / L) v_new);  wrap(u) = u - floor(u), 1 -> 0
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
                    Ex = Ex + w * e[0];
                    Ey = Ey + w * e[1];
                    Ez = Ez + w * e[2];
                }
        const REAL ax = hc * Ex, ay = hc * Ey, az = hc * Ez;
        REAL ux = vx[p] + ax, uy = vy[p] + ay, uz = vz[p] + az;
        if (has_b) {
            const REAL px = ux + (uy * tz - uz * ty);
            const REAL py = uy + (uz * tx - ux * tz);
            const REAL pz = uz + (ux * ty - uy * tx);
            const REAL qx = ux + (py * sz - pz * sy);
            const REAL qy = uy + (pz * sx - px * sz);
            const REAL qz = uz + (px * sy - py * sx);
            ux = qx; uy = qy; uz = qz;
        }
        const REAL nvx = ux + ax, nvy = uy + ay, nvz = uz + az;
        vx[p] = nvx; vy[p] = nvy; vz[p] = nvz;
        x[p] = FN(es3d_wrap)(x[p] + dx * nvx);
        y[p] = FN(es3d_wrap)(y[p] + dy * nvy);
        z[p] = FN(es3d_wrap)(z[p] + dz * nvz);
    }
}

#undef FN
#undef CAT
#undef CAT_
