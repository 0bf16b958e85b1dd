/*
 * pic_oracle_impl.h — body of the CPU oracle, included twice by pic_oracle.c with
 *   REAL = float  / FN(x) = orc_f32_##x
 *   REAL = double / FN(x) = orc_f64_##x
 *
 * TEST INFRASTRUCTURE ONLY (see pic_oracle.h).  Every function is a restatement of
 * one GLSL program or host routine of the reference, cited as file:line under
 * /root/reference/public/javascripts/.  Arrays are RGBA "textures" of REAL, laid
 * out exactly as the reference's Float32Array textures: texel (i,j) of a W-wide
 * texture is at 4*(i + j*W).  Expression order follows the GLSL text left to
 * right; the file is compiled with -ffp-contract=off so no product-sum is fused.
 */

/* NEAREST + CLAMP_TO_EDGE lookup (utilities.js:528-531): texel = clamp(floor(u*W), 0, W-1).
 * Convention for inputs GL leaves undefined: NaN selects texel 0. */
static inline int FN(ngp)(REAL u, int W)
{
    REAL t = u * (REAL)W;
    if (!(t >= (REAL)0)) return 0;
    if (t >= (REAL)W) return W - 1;
    return (int)t;
}

/* programStepRandA/B (empic.js:783-820, :858-895), K3. */
void FN(step_rand)(const REAL* rand_in, const REAL* entropy, REAL* rand_out, size_t n)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n; ++p) {
        const REAL* rd = rand_in + 4 * p;
        REAL x0 = rd[2], x1 = rd[3];
        int ei = FN(ngp)(x0, ORC_N_ENTROPY), ej = FN(ngp)(x1, ORC_N_ENTROPY);
        const REAL* s = entropy + 4 * ((size_t)ei + (size_t)ORC_N_ENTROPY * ej);
        x0 = (REAL)0.999 * x0 + (REAL)0.001 * s[2];
        x1 = (REAL)0.999 * x1 + (REAL)0.001 * s[3];
        REAL m0 = rd[0] + s[0], m1 = rd[1] + s[1];
        REAL* o = rand_out + 4 * p;
        o[0] = (m0 > (REAL)1) ? m0 - (REAL)1 : m0;
        o[1] = (m1 > (REAL)1) ? m1 - (REAL)1 : m1;
        o[2] = (REAL)4 * x0 * ((REAL)1 - x0);
        o[3] = (REAL)4 * x1 * ((REAL)1 - x1);
    }
}

/* step_velocity_frag (empic.js:729-778), K1. */
void FN(step_velocity)(const REAL* pos, const REAL* vel, const REAL* rnd,
                       const REAL* R1, const REAL* R2, const REAL* R3, const REAL* A,
                       int nr, int nz, REAL* vel_out, size_t n)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        const REAL* V = vel + 4 * p;
        const REAL* rd = rnd + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        REAL dx = P[0] / r, dy = P[1] / r;
        REAL vr = V[0] * dx + V[1] * dy;
        REAL va = V[1] * dx - V[0] * dy;
        REAL vz = V[2];
        size_t c = 4 * ((size_t)FN(ngp)(r, nr) + (size_t)nr * FN(ngp)(P[2], nz));
        REAL cx = ((R1[c] * vr + R1[c + 1] * va) + R1[c + 2] * vz) + A[c];
        REAL cy = ((R2[c] * vr + R2[c + 1] * va) + R2[c + 2] * vz) + A[c + 1];
        REAL cz = ((R3[c] * vr + R3[c + 1] * va) + R3[c + 2] * vz) + A[c + 2];
        REAL* o = vel_out + 4 * p;
        if (P[3] > (REAL)0.5) {
            o[0] = cx * dx - cy * dy;
            o[1] = cx * dy + cy * dx;
            o[2] = cz;
            o[3] = (REAL)1;
        } else { /* just re-injected: new thermal velocity (empic.js:772, quirk Q4) */
            o[0] = (REAL)0.001 * ((REAL)2 * rd[0] - (REAL)1);
            o[1] = (REAL)0.001 * ((REAL)2 * rd[1] - (REAL)1);
            o[2] = (REAL)0.001 * ((REAL)2 * rd[2] - (REAL)1);
            o[3] = (REAL)0.001 * (REAL)1;
        }
    }
}

/* step_position_frag (empic.js:692-726), K2. */
void FN(step_position)(const REAL* pos, const REAL* vel_new, const REAL* rnd,
                       const REAL* sink, const REAL* inv_cdf, int nr, int nz,
                       REAL step_factor, REAL* pos_out, size_t n)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        const REAL* V = vel_new + 4 * p;
        const REAL* rd = rnd + 4 * p;
        REAL nx = P[0] + step_factor * V[0];
        REAL ny = P[1] + step_factor * V[1];
        REAL nzp = P[2] + step_factor * V[2];
        REAL r = SQRT(nx * nx + ny * ny);
        size_t q = 4 * ((size_t)FN(ngp)(rd[0], ORC_N_CDF) + (size_t)ORC_N_CDF * FN(ngp)(rd[1], ORC_N_CDF));
        size_t c = 4 * ((size_t)FN(ngp)(r, nr) + (size_t)nr * FN(ngp)(nzp, nz));
        REAL* o = pos_out + 4 * p;
        if (sink[c] > (REAL)0.5) {
            o[0] = nx; o[1] = ny; o[2] = nzp; o[3] = (REAL)1;
        } else { /* lost: re-inject on the axis plane y = 0 (empic.js:719) */
            o[0] = inv_cdf[q]; o[1] = (REAL)0; o[2] = inv_cdf[q + 1]; o[3] = (REAL)0;
        }
    }
}

/* out.step (empic.js:1436-1469) ncalls times: RandB, VelB, PosB, RandA, VelA, PosA.
 * Buffers *_A hold the state on entry and on exit; *_B are scratch of equal size.
 * Bindings per pass: empic.js:815-853, :890-928 (a velocity/position pass reads the
 * OLD rand of its sub-step; the position pass reads the NEW velocity). */
void FN(step)(REAL* pos_A, REAL* vel_A, REAL* rand_A, REAL* pos_B, REAL* vel_B, REAL* rand_B,
              const REAL* entropy, const REAL* R1, const REAL* R2, const REAL* R3, const REAL* A,
              const REAL* sink, const REAL* inv_cdf, int nr, int nz, REAL step_factor,
              size_t n, int ncalls)
{
    for (int k = 0; k < ncalls; ++k) {
        FN(step_rand)(rand_A, entropy, rand_B, n);
        FN(step_velocity)(pos_A, vel_A, rand_A, R1, R2, R3, A, nr, nz, vel_B, n);
        FN(step_position)(pos_A, vel_B, rand_A, sink, inv_cdf, nr, nz, step_factor, pos_B, n);
        FN(step_rand)(rand_B, entropy, rand_A, n);
        FN(step_velocity)(pos_B, vel_B, rand_B, R1, R2, R3, A, nr, nz, vel_A, n);
        FN(step_position)(pos_B, vel_A, rand_B, sink, inv_cdf, nr, nz, step_factor, pos_A, n);
    }
}

/* NGP cell i + j*nr the velocity pass gathers from (integer parity output). */
void FN(cells)(const REAL* pos, size_t n, int nr, int nz, int32_t* cells)
{
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        cells[p] = FN(ngp)(r, nr) + nr * FN(ngp)(P[2], nz);
    }
}

/* programPre1/2/3 + programPreA (empic.js:506-659), K8/K9.  h is the uniform u_h;
 * f_rz = factor_r/factor_z and f_zr = factor_z/factor_r are the literals baked into
 * the shader text (empic.js:527, :566, :606); physical_a = 0 keeps quirk Q1
 * (scalar u_h*dot(E,B) added to every component, empic.js:645). */
void FN(precalc)(const REAL* B, const REAL* E, int nr, int nz, REAL h,
                 REAL factor_r, REAL factor_z, REAL f_rz, REAL f_zr,
                 REAL* R1, REAL* R2, REAL* R3, REAL* A, int physical_a)
{
    size_t ncell = (size_t)nr * nz;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t c = 0; c < ncell; ++c) {
        REAL Bx = B[4 * c], By = B[4 * c + 1], Bz = B[4 * c + 2];
        REAL Ex = E[4 * c], Ey = E[4 * c + 1], Ez = E[4 * c + 2];
        REAL Bmag = SQRT((Bx * Bx + By * By) + Bz * Bz);
        REAL hB2 = h * h * Bmag * Bmag;
        REAL factor = (REAL)2 / ((REAL)1 + hB2);
        REAL diag = (REAL)1 - hB2 * factor;
        REAL fh = factor * h;

        R1[4 * c + 0] = diag + fh * h * Bx * Bx;
        R1[4 * c + 1] = fh * (Bz + h * Bx * By);
        R1[4 * c + 2] = (fh * (-By + h * Bx * Bz)) * f_rz;
        R1[4 * c + 3] = (REAL)1;

        R2[4 * c + 0] = fh * (-Bz + h * By * Bx);
        R2[4 * c + 1] = diag + fh * h * By * By;
        R2[4 * c + 2] = (fh * (Bx + h * By * Bz)) * f_rz;
        R2[4 * c + 3] = (REAL)1;

        R3[4 * c + 0] = (fh * (By + h * Bz * Bx)) * f_zr;
        R3[4 * c + 1] = (fh * (-Bx + h * Bz * By)) * f_zr;
        R3[4 * c + 2] = diag + fh * h * Bz * Bz;
        R3[4 * c + 3] = (REAL)1;

        REAL a = h * ((REAL)2 - hB2 * factor);
        REAL b = h * h * factor;
        REAL cx = Ey * Bz - Ez * By;
        REAL cy = Ez * Bx - Ex * Bz;
        REAL cz = Ex * By - Ey * Bx;
        REAL dot = (Ex * Bx + Ey * By) + Ez * Bz;
        REAL hd = h * dot;
        REAL Ax, Ay, Az;
        if (physical_a) {
            Ax = (a * Ex + b * (cx + hd * Bx)) / (REAL)2.998e8;
            Ay = (a * Ey + b * (cy + hd * By)) / (REAL)2.998e8;
            Az = (a * Ez + b * (cz + hd * Bz)) / (REAL)2.998e8;
        } else {
            Ax = (a * Ex + b * (cx + hd)) / (REAL)2.998e8;
            Ay = (a * Ey + b * (cy + hd)) / (REAL)2.998e8;
            Az = (a * Ez + b * (cz + hd)) / (REAL)2.998e8;
        }
        A[4 * c + 0] = Ax * factor_r;
        A[4 * c + 1] = Ay * factor_r;
        A[4 * c + 2] = Az * factor_z;
        A[4 * c + 3] = (REAL)1;
    }
}

/* programMoments01 (empic.js:980-1035) drawn as n points with blend ONE,ONE after a
 * clear to 0 (empic.js:1473-1478), K4.  The point sprite is 11x11 pixels
 * (u_pointsize = nshape); for a particle whose window position is (ic+f, jc+g),
 * 0<f,g<1, the covered pixel centres are ic-5..ic+5 / jc-5..jc+5 and the NEAREST
 * lookup of gl_PointCoord selects stamp texel (di+5, 5-dj) (gl_PointCoord.t runs
 * downwards); the stamp is symmetric so the flip is invisible.  A point whose
 * centre lies outside the clip volume is discarded whole, footprints are cropped
 * at the target's edges.  stamp is the red channel, 121 floats (empic.js:949-971).
 * moments is cleared here.  Accumulation is serial in particle order. */
void FN(deposit)(const REAL* pos, const REAL* vel, size_t n, const float* stamp,
                 int nr, int nz, REAL* moments)
{
    size_t ncell = (size_t)nr * nz;
    for (size_t c = 0; c < 4 * ncell; ++c) moments[c] = (REAL)0;
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        const REAL* V = vel + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        REAL z = P[2];
        if (!(r >= (REAL)0 && r <= (REAL)1 && z >= (REAL)0 && z <= (REAL)1)) continue;
        REAL dx = P[0] / r, dy = P[1] / r;
        REAL vr = V[0] * dx + V[1] * dy;
        REAL va = V[1] * dx - V[0] * dy;
        REAL col[4] = { (REAL)0.001 * vr, (REAL)0.001 * va, (REAL)0.001 * V[2], (REAL)0.001 * (REAL)1 };
        int ic = (int)(r * (REAL)nr), jc = (int)(z * (REAL)nz);
        for (int dj = -5; dj <= 5; ++dj) {
            int j = jc + dj;
            if (j < 0 || j >= nz) continue;
            for (int di = -5; di <= 5; ++di) {
                int i = ic + di;
                if (i < 0 || i >= nr) continue;
                REAL w = (REAL)stamp[(di + 5) + 11 * (5 - dj)];
                REAL* m = moments + 4 * ((size_t)i + (size_t)nr * j);
                m[0] += col[0] * w;
                m[1] += col[1] * w;
                m[2] += col[2] * w;
                m[3] += col[3] * w;
            }
        }
    }
}

/* The same draw as a REAL rasteriser executes it (tests/golden/webgl_*: the reference run by Chromium's WebGL 1 on
 * ANGLE/SwiftShader, oracle/make_golden_webgl.py).  deposit() above assumes window coordinates of infinite precision and
 * the whole-point clipping of the GL ES 2.0 text; a rasteriser does neither.  What SwiftShader does, and what this
 * function restates (subpixel_bits = b, SwiftShader: 4 = gl.getParameter(SUBPIXEL_BITS)):
 *   - clip coordinates (2r-1, 2z-1) in float (empic.js:997);
 *   - viewport transform to a fixed-point window position in pixel-centre coordinates (pixel p's centre at p*2^b), y
 *     running DOWNWARDS: X = rint(X0 + ndc.x*Wb), Y = rint(Y0 + ndc.y*(-Hb)) with Wb = (nr/2)*2^b, X0 = Wb - 2^(b-1)
 *     (same with nz), one float rounding per operation, rint = round half to even;
 *   - the sprite is the square X +- 11*2^(b-1); a pixel is covered when its centre lies inside, left/top edges
 *     inclusive: columns p0 .. p0+10 with p0 = ceil((X - 11*2^(b-1)) / 2^b), rows likewise counted from the top;
 *     gl_PointCoord's NEAREST texel of covered pixel k of a row or column is k (observed: exact);
 *   - a point is NOT discarded when its centre leaves the clip volume: the square is clipped, i.e. the footprint is
 *     cropped at the target's edges whatever the centre is (centres up to 5.5 pixels outside still deposit); only
 *     non-finite positions are dropped;
 *   - blending in particle order.
 * A particle whose window coordinate lies within 2^-(b+1) pixels above a pixel edge therefore lands one cell lower
 * than under deposit()'s convention (1/16 of the particles per axis at b = 4, 1/256 at b = 8).  Not modelled:
 * SwiftShader flushes denormal products to zero and, on a measure-zero set, covers a 12th row/column whose texels
 * repeat the stamp's outermost ring (values <= 1.7e-34): fixtures agree to 1e-33 of the image's maximum, bit for bit
 * above that. */
static inline int FN(raster_origin)(REAL u, int W, int bits, int y_down, int* first)
{
    REAL ndc = (REAL)2 * u - (REAL)1;
    REAL wb = (REAL)W * (REAL)0.5 * (REAL)(1 << bits);
    REAL x0 = wb - (REAL)(1 << bits) * (REAL)0.5;
    REAL t = ndc * (y_down ? -wb : wb);
    REAL s = x0 + t;
    if (!(s > (REAL)-1073741824.0 && s < (REAL)1073741824.0)) return 0; /* NaN, infinite or absurd: dropped */
    long X = lrint((double)s); /* round half to even (the default rounding mode), as cvtps2dq does */
    long half = 11L * (1L << bits) / 2;
    long a = X - half, sc = 1L << bits;
    long p0 = (a >= 0) ? (a + sc - 1) / sc : -((-a) / sc); /* ceil(a / sc) */
    *first = y_down ? (int)(W - 1 - (p0 + 10)) : (int)p0;
    return 1;
}

void FN(deposit_raster)(const REAL* pos, const REAL* vel, size_t n, const float* stamp,
                        int nr, int nz, REAL* moments, int subpixel_bits)
{
    size_t ncell = (size_t)nr * nz;
    for (size_t c = 0; c < 4 * ncell; ++c) moments[c] = (REAL)0;
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        const REAL* V = vel + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        REAL z = P[2];
        int i0, j0;
        if (!FN(raster_origin)(r, nr, subpixel_bits, 0, &i0) || !FN(raster_origin)(z, nz, subpixel_bits, 1, &j0)) continue;
        if (i0 >= nr || i0 + 10 < 0 || j0 >= nz || j0 + 10 < 0) continue;
        REAL dx = P[0] / r, dy = P[1] / r;
        REAL vr = V[0] * dx + V[1] * dy;
        REAL va = V[1] * dx - V[0] * dy;
        REAL col[4] = { (REAL)0.001 * vr, (REAL)0.001 * va, (REAL)0.001 * V[2], (REAL)0.001 * (REAL)1 };
        for (int dj = 0; dj < 11; ++dj) {
            int j = j0 + dj;
            if (j < 0 || j >= nz) continue;
            for (int di = 0; di < 11; ++di) {
                int i = i0 + di;
                if (i < 0 || i >= nr) continue;
                REAL w = (REAL)stamp[di + 11 * (10 - dj)];
                REAL* m = moments + 4 * ((size_t)i + (size_t)nr * j);
                m[0] += col[0] * w;
                m[1] += col[1] * w;
                m[2] += col[2] * w;
                m[3] += col[3] * w;
            }
        }
    }
}

/* Sprite-centre cells of deposit_raster(): (first column + 5, first row + 5); they may lie up to 6 cells outside the
 * grid.  ci = INT32_MIN for a dropped point. */
void FN(raster_cells)(const REAL* pos, size_t n, int nr, int nz, int subpixel_bits, int32_t* ci, int32_t* cj)
{
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        int i0, j0;
        if (!FN(raster_origin)(r, nr, subpixel_bits, 0, &i0) || !FN(raster_origin)(P[2], nz, subpixel_bits, 1, &j0)) {
            ci[p] = INT32_MIN; cj[p] = INT32_MIN;
            continue;
        }
        ci[p] = i0 + 5; cj[p] = j0 + 5;
    }
}

/* EXTENSION (SURVEY.md 8(b) key shape:'cic'; no reference counterpart, parity unpinned): the same vertex colour
 * spread bilinearly over the four cell CENTRES around the point instead of the 11x11 stamp.  With the window
 * coordinates (gi, gj) = (r*nr, z*nz) of empic.js:997-999: i0 = floor(gi - 0.5), f = gi - 0.5 - i0, weights
 * (1-f, f) on cells i0, i0+1 (same along z), w = wr*wz, colour*w accumulated in particle order.  Clipping of
 * points outside [0,1]^2 and cropping at the grid's edges as for the sprite. */
void FN(deposit_cic)(const REAL* pos, const REAL* vel, size_t n, int nr, int nz, REAL* moments)
{
    size_t ncell = (size_t)nr * nz;
    for (size_t c = 0; c < 4 * ncell; ++c) moments[c] = (REAL)0;
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        const REAL* V = vel + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        REAL z = P[2];
        if (!(r >= (REAL)0 && r <= (REAL)1 && z >= (REAL)0 && z <= (REAL)1)) continue;
        REAL dx = P[0] / r, dy = P[1] / r;
        REAL vr = V[0] * dx + V[1] * dy;
        REAL va = V[1] * dx - V[0] * dy;
        REAL col[4] = { (REAL)0.001 * vr, (REAL)0.001 * va, (REAL)0.001 * V[2], (REAL)0.001 * (REAL)1 };
        REAL gi = r * (REAL)nr - (REAL)0.5, gj = z * (REAL)nz - (REAL)0.5;
        REAL fi0 = FLOOR(gi), fj0 = FLOOR(gj);
        int i0 = (int)fi0, j0 = (int)fj0;
        REAL wr[2], wz[2];
        wr[1] = gi - fi0; wr[0] = (REAL)1 - wr[1];
        wz[1] = gj - fj0; wz[0] = (REAL)1 - wz[1];
        for (int b = 0; b < 2; ++b) {
            int j = j0 + b;
            if (j < 0 || j >= nz) continue;
            for (int a = 0; a < 2; ++a) {
                int i = i0 + a;
                if (i < 0 || i >= nr) continue;
                REAL w = wr[a] * wz[b];
                REAL* m = moments + 4 * ((size_t)i + (size_t)nr * j);
                m[0] += col[0] * w;
                m[1] += col[1] * w;
                m[2] += col[2] * w;
                m[3] += col[3] * w;
            }
        }
    }
}

#ifdef _OPENMP
/* Timing variant for bench.py's all-cores baseline (built only into libpic_oracle_omp.so):
 * the same per-particle arithmetic as deposit(), particles split into contiguous ranges, one
 * private target per thread, targets added in thread order.  The summation order differs from
 * the serial one, so parity tests never use it. */
void FN(deposit_threads)(const REAL* pos, const REAL* vel, size_t n, const float* stamp,
                         int nr, int nz, REAL* moments)
{
    size_t ncell = (size_t)nr * nz;
    int nt = omp_get_max_threads();
    REAL* priv = (REAL*)malloc(sizeof(REAL) * 4 * ncell * (size_t)nt);
    if (!priv) { FN(deposit)(pos, vel, n, stamp, nr, nz, moments); return; }
#pragma omp parallel num_threads(nt)
    {
        int t = omp_get_thread_num();
        size_t b = n * (size_t)t / (size_t)nt, e = n * (size_t)(t + 1) / (size_t)nt;
        FN(deposit)(pos + 4 * b, vel + 4 * b, e - b, stamp, nr, nz, priv + 4 * ncell * (size_t)t);
    }
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < 4 * ncell; ++c) {
        REAL acc = priv[c];
        for (int t = 1; t < nt; ++t) acc += priv[c + 4 * ncell * (size_t)t];
        moments[c] = acc;
    }
    free(priv);
}
#endif

/* Per-particle deposit cell (ic + (nr+1)*jc, or -1 when the point is clipped). */
void FN(deposit_cells)(const REAL* pos, size_t n, int nr, int nz, int32_t* cells)
{
    for (size_t p = 0; p < n; ++p) {
        const REAL* P = pos + 4 * p;
        REAL r = SQRT(P[0] * P[0] + P[1] * P[1]);
        REAL z = P[2];
        if (!(r >= (REAL)0 && r <= (REAL)1 && z >= (REAL)0 && z <= (REAL)1)) { cells[p] = -1; continue; }
        cells[p] = (int)(r * (REAL)nr) + (nr + 1) * (int)(z * (REAL)nz);
    }
}

/* programNormalizeMoments01 (empic.js:1042-1066), K5; v_texCoord.x = (i+0.5)/nr. */
void FN(normalise)(const REAL* moments, int nr, int nz, REAL* norm)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr; ++i) {
            size_t c = 4 * ((size_t)i + (size_t)nr * j);
            REAL x = ((REAL)i + (REAL)0.5) / (REAL)nr;
            REAL a = moments[c + 3];
            REAL m[4] = { 0, 0, 0, 0 };
            if (a > (REAL)0) {
                m[0] = moments[c] / a; m[1] = moments[c + 1] / a; m[2] = moments[c + 2] / a; m[3] = a;
            }
            for (int k = 0; k < 4; ++k) norm[c + k] = (REAL)1000 * m[k] * (REAL)0.5 / x;
        }
}

/* programAvgMoments / avg_frag (empic.js:262-282, :1075-1084), K6, u_ratio = 0.01;
 * followed by the programSet copy avgA -> avgB (empic.js:1490-1495), K7. */
void FN(avg)(const REAL* next, REAL* avg_B, REAL* avg_A, REAL ratio, size_t ncell)
{
    REAL keep = (REAL)1 - ratio;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (size_t c = 0; c < 4 * ncell; ++c) {
        avg_A[c] = ratio * next[c] + keep * avg_B[c];
        avg_B[c] = avg_A[c];
    }
}

/* out.set({position|velocity}) host loop (empic.js:1199-1244): value*factor in
 * double, one rounding into the texture's element type, w = 1. */
void FN(normalise_particles)(const double* aos3, size_t n, double factor_xy, double factor_z, REAL* out4)
{
    for (size_t p = 0; p < n; ++p) {
        out4[4 * p + 0] = (REAL)(aos3[3 * p + 0] * factor_xy);
        out4[4 * p + 1] = (REAL)(aos3[3 * p + 1] * factor_xy);
        out4[4 * p + 2] = (REAL)(aos3[3 * p + 2] * factor_z);
        out4[4 * p + 3] = (REAL)1;
    }
}

/* out.set({E|B}) packing (empic.js:1159-1197): value[i][j][k] -> 4*(i + j*nr) + k, w = 1. */
void FN(pack_grid3)(const double* in_ijk, int nr, int nz, REAL* rgba)
{
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nz; ++j) {
            size_t c = 4 * ((size_t)i + (size_t)nr * j);
            const double* v = in_ijk + 3 * ((size_t)i * nz + j);
            rgba[c] = (REAL)v[0]; rgba[c + 1] = (REAL)v[1]; rgba[c + 2] = (REAL)v[2]; rgba[c + 3] = (REAL)1;
        }
}

/* out.set({sink_mask}) packing (empic.js:1246-1260): red channel only (quirk Q9). */
void FN(pack_sink)(const double* in_ij, int nr, int nz, REAL* rgba)
{
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nz; ++j)
            rgba[4 * ((size_t)i + (size_t)nr * j)] = (REAL)in_ij[(size_t)i * nz + j];
}

/* programCurrentLoopShape (empic.js:295-345), K10: Biot-Savart sum over 1000
 * segments of a half circle for a unit loop of normalised radius u_R, evaluated at
 * texel centres ((i+.5)/nr, (j+.5)/nz).  out is RGBA (Bx, 0, Bz, 1). */
void FN(loop_shape)(REAL u_R, int nr, int nz, REAL* out)
{
    const REAL pi = (REAL)3.14159265359;
    REAL constant = u_R * (REAL)0.001 * (REAL)1.25663706e-6 / ((REAL)4.0 * pi);
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr; ++i) {
            REAL tx = ((REAL)i + (REAL)0.5) / (REAL)nr;
            REAL ty = ((REAL)j + (REAL)0.5) / (REAL)nz;
            REAL Bx = 0, Bz = 0;
            for (int k = 0; k < 1000; ++k) {
                REAL cosine = COS(pi * ((REAL)k + (REAL)0.5) / (REAL)1000.0);
                REAL r = SQRT(u_R * u_R + tx * tx + ty * ty - (REAL)2.0 * tx * u_R * cosine);
                REAL factor = (r > (REAL)0) ? constant * (REAL)1.0 / (r * r * r) : (REAL)0;
                Bx += ty * factor * cosine;
                Bz += factor * (u_R - tx * cosine);
            }
            REAL* o = out + 4 * ((size_t)i + (size_t)nr * j);
            o[0] = Bx; o[1] = (REAL)0; o[2] = Bz; o[3] = (REAL)1;
        }
}

/* programCurrentLoop (empic.js:349-389) drawn into B with blend ONE,ONE
 * (empic.js:1352-1363), K11.  u_R = r*factor_r, u_Z = z*factor_z.  Quirk Q6: the
 * far-field test is `b > 2.0` without abs. */
void FN(add_current_loop)(REAL* B, const REAL* shape_half, const REAL* shape_tenth, int nr, int nz,
                          REAL u_R, REAL u_Z, REAL u_I)
{
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr; ++i) {
            REAL tx = ((REAL)i + (REAL)0.5) / (REAL)nr;
            REAL ty = ((REAL)j + (REAL)0.5) / (REAL)nz;
            REAL a = tx / u_R;
            REAL b = (ty - u_Z) / u_R;
            REAL sgn = (b > (REAL)0) ? (REAL)1 : ((b < (REAL)0) ? (REAL)-1 : (REAL)0);
            REAL ab = (b < (REAL)0) ? -b : b;
            const REAL* t;
            if (a > (REAL)2.0 || b > (REAL)2.0)
                t = shape_tenth + 4 * ((size_t)FN(ngp)(a / (REAL)10.0, nr) + (size_t)nr * FN(ngp)(ab / (REAL)10.0, nz));
            else
                t = shape_half + 4 * ((size_t)FN(ngp)(a / (REAL)2.0, nr) + (size_t)nr * FN(ngp)(ab / (REAL)2.0, nz));
            REAL* o = B + 4 * ((size_t)i + (size_t)nr * j);
            o[0] += (u_I * sgn) * t[0];
            o[1] += (u_I * (REAL)1) * t[1];
            o[2] += (u_I * (REAL)1) * t[2];
            o[3] += (u_I * (REAL)1) * t[3];
        }
}

/* programCurrentZ / programBZ / programBTheta (empic.js:392-464) drawn into B with
 * blend ONE,ONE (empic.js:1380-1411), K12.  Quirk Q7: the shaders read-modify an
 * undefined gl_FragColor; it is taken as 0.  kind: 0 = line current on the axis,
 * 1 = uniform Bz, 2 = uniform Btheta. */
void FN(add_uniform)(REAL* B, int nr, int nz, int kind, REAL value)
{
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nr; ++i) {
            REAL* o = B + 4 * ((size_t)i + (size_t)nr * j);
            if (kind == 0) {
                REAL tx = ((REAL)i + (REAL)0.5) / (REAL)nr;
                o[1] += value * (REAL)1.25663706e-6 / ((REAL)2.0 * (REAL)3.14159265359 * tx);
            } else if (kind == 1) {
                o[2] += value;
            } else {
                o[1] += value;
            }
            o[3] += (REAL)1;
        }
}

/* ---- counter-based RNG mode (an EXTENSION, SURVEY.md 8(d): "RNG: counter-based
 * Philox-4x32-10 ... stream = particle id"; the reference's own generator is the
 * entropy-table walk above).  The random vector of particle `id` at sub-step `t` is
 * Philox4x32-10(counter = (id, t_lo, t_hi, 0x5EED), key = seed), each word mapped to
 * (w >> 8) * 2^-24.  It plays the role of the rand texel in both passes of a sub-step:
 * .xyz re-seeds the velocity of a just re-injected particle (empic.js:772), .xy picks
 * the re-injection point (empic.js:717-719).  Nothing is stored per particle. */
void FN(step_counter)(REAL* pos_A, REAL* vel_A, REAL* pos_B, REAL* vel_B,
                      const REAL* R1, const REAL* R2, const REAL* R3, const REAL* A,
                      const REAL* sink, const REAL* inv_cdf, int nr, int nz, REAL step_factor,
                      size_t n, int ncalls, uint64_t seed, uint64_t t0)
{
    REAL* rnd = (REAL*)malloc(sizeof(REAL) * 4 * n);
    uint64_t t = t0;
    for (int k = 0; k < 2 * ncalls; ++k, ++t) {
        for (size_t p = 0; p < n; ++p) {
            uint32_t w[4];
            orc_philox4x32_10((uint32_t)p, (uint32_t)t, (uint32_t)(t >> 32), 0x5EEDu, (uint32_t)seed, (uint32_t)(seed >> 32), w);
            for (int c = 0; c < 4; ++c) rnd[4 * p + c] = (REAL)(w[c] >> 8) * (REAL)(1.0 / 16777216.0);
        }
        if ((k & 1) == 0) {
            FN(step_velocity)(pos_A, vel_A, rnd, R1, R2, R3, A, nr, nz, vel_B, n);
            FN(step_position)(pos_A, vel_B, rnd, sink, inv_cdf, nr, nz, step_factor, pos_B, n);
        } else {
            FN(step_velocity)(pos_B, vel_B, rnd, R1, R2, R3, A, nr, nz, vel_A, n);
            FN(step_position)(pos_B, vel_A, rnd, sink, inv_cdf, nr, nz, step_factor, pos_A, n);
        }
    }
    free(rnd);
}
