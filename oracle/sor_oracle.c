/*
 * sor_oracle.c — CPU oracle for the reference's dense iterative solver
 * (matrix_webgl.js makeSORIterative).
 *
 * TEST INFRASTRUCTURE ONLY, like pic_oracle.c: loaded by tests/ (and smoke/bench as the
 * checker), never by libfusionpic.so.  Each function restates one GLSL program or host
 * routine, cited as matrix_webgl.js:line, on arrays laid out exactly like the reference's
 * RGBA float textures (texel (i,j) of a W-wide texture at 4*(i + W*j)).  Pinned by
 * tests/golden/swgl_sor.* (the reference's own host code and shader strings evaluated in
 * software, oracle/make_golden.js section 9): every texture and every returned number is
 * reproduced bit for bit (tests/test_oracle_sor.py).
 *
 * Sizes (matrix_webgl.js:44-52): vh = 2^n_power, vectors are vh x vh RGBA = L = 4 vh^2
 * elements, element e = 4*(X + vh*Y) + channel; the iteration matrix is an mh x mh RGBA
 * texture, mh = 2 vh^2, made of (2vh) x (2vh) blocks of vh x vh texels: block (bx,by) holds
 * matrix row bx + 2vh*by, texel (vx,vy) of a block holds columns 4*(vx + vh*vy) + channel.
 * All sizes are powers of two, so every texture coordinate the shaders compute is exact and
 * the lookups below are plain integer indexing.
 *
 * Build: oracle/Makefile (-ffp-contract=off: no product-sum is fused).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

double orc_tofixed20(double x); /* pic_oracle.c: N(x) = x.toFixed(20) re-read */

static int vh_of(int n_power) { return 1 << n_power; }

/* Math.max: NaN if either argument is NaN */
static double js_max(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? b : a); }

/* programR (matrix_webgl.js:222-262): R[row][col] = row == col ? 0 : -A[row][col] / A[row][row],
 * times the literal N(omega) when omega != 1.  A is row-major, A[col + L*row] (set_matrix, :458-466). */
void orc_sor_build_R(const float* A, int n_power, double omega, float* R)
{
    const int vh = vh_of(n_power), mh = 2 * vh * vh, L = 4 * vh * vh;
    const float w = (float)orc_tofixed20(omega);
    for (int ny = 0; ny < mh; ++ny)
        for (int nx = 0; nx < mh; ++nx) {
            const int row = nx / vh + 2 * vh * (ny / vh);
            const int col = 4 * (nx % vh + vh * (ny % vh));
            float* o = R + 4 * ((size_t)nx + (size_t)mh * ny);
            for (int k = 0; k < 4; ++k) {
                float v = (row == col + k) ? 0.0f : -A[(size_t)(col + k) + (size_t)L * row] / A[(size_t)row + (size_t)L * row];
                o[k] = (omega != 1.0) ? w * v : v;
            }
        }
}

/* programC (matrix_webgl.js:266-301): C[e] = b[e] / A[e][e], times N(omega) when omega != 1. */
void orc_sor_build_C(const float* A, const float* b, int n_power, double omega, float* C)
{
    const int vh = vh_of(n_power), L = 4 * vh * vh;
    const float w = (float)orc_tofixed20(omega);
    for (int e = 0; e < L; ++e) {
        float v = b[e] / A[(size_t)e + (size_t)L * e];
        C[e] = (omega != 1.0) ? w * v : v;
    }
}

/* out.mv_product (matrix_webgl.js:535-558): programMVproduct (:305-330), the n_power
 * pair-summing passes (:341-385) and programResult (:389-424).
 *   product texel = R texel * x_guess texel of the block-local position, per channel;
 *   each pass sums 2x2 texels in the order (+x+y) + (-x+y) + (+x-y) + (-x-y);
 *   result texel (X,Y) = (dot(S(2X,2Y),1), dot(S(2X+1,2Y),1), dot(S(2X,2Y+1),1), dot(S(2X+1,2Y+1),1))
 *                        + C + N(1-omega) * x_guess   (last term only when omega != 1),
 * where S is the fully summed (2vh) x (2vh) texture, S(bx,by) = matrix row bx + 2vh*by.
 * NOTE (kept, quirk Q14 in DESIGN.md): element e = 4(X+vh*Y)+c therefore receives matrix row
 * (2X + c%2) + 2vh*(2Y + c/2), which equals e only when vh = 1. */
void orc_sor_mv_product(const float* R, const float* C, const float* x_guess, int n_power, double omega, float* x_out)
{
    const int vh = vh_of(n_power), mh = 2 * vh * vh;
    float* cur = (float*)malloc(sizeof(float) * 4 * (size_t)mh * mh);
    float* nxt = (float*)malloc(sizeof(float) * (size_t)mh * mh);
    for (int ny = 0; ny < mh; ++ny)
        for (int nx = 0; nx < mh; ++nx) {
            const float* m = R + 4 * ((size_t)nx + (size_t)mh * ny);
            const float* v = x_guess + 4 * ((nx % vh) + vh * (ny % vh));
            float* o = cur + 4 * ((size_t)nx + (size_t)mh * ny);
            for (int k = 0; k < 4; ++k) o[k] = m[k] * v[k];
        }
    int w = mh;
    for (int pass = 0; pass < n_power; ++pass) {
        const int h = w / 2;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < h; ++x)
                for (int k = 0; k < 4; ++k) {
                    const float pp = cur[4 * ((size_t)(2 * x + 1) + (size_t)w * (2 * y + 1)) + k];
                    const float mp = cur[4 * ((size_t)(2 * x) + (size_t)w * (2 * y + 1)) + k];
                    const float pm = cur[4 * ((size_t)(2 * x + 1) + (size_t)w * (2 * y)) + k];
                    const float mm = cur[4 * ((size_t)(2 * x) + (size_t)w * (2 * y)) + k];
                    nxt[4 * ((size_t)x + (size_t)h * y) + k] = ((pp + mp) + pm) + mm;
                }
        memcpy(cur, nxt, sizeof(float) * 4 * (size_t)h * h);
        w = h;
    }
    /* w == 2*vh */
    const float keep = (float)orc_tofixed20(1.0 - omega);
    for (int Y = 0; Y < vh; ++Y)
        for (int X = 0; X < vh; ++X)
            for (int k = 0; k < 4; ++k) {
                const float* s = cur + 4 * ((size_t)(2 * X + (k & 1)) + (size_t)w * (2 * Y + (k >> 1)));
                const float sum = ((s[0] * 1.0f + s[1] * 1.0f) + s[2] * 1.0f) + s[3] * 1.0f;
                const size_t e = 4 * ((size_t)X + (size_t)vh * Y) + k;
                float r = sum + C[e];
                if (omega != 1.0) r = r + keep * x_guess[e];
                x_out[e] = r;
            }
    free(cur);
    free(nxt);
}

/* programStats (matrix_webgl.js:428-452): per vector texel
 * (dot(x1,x2)*0.25, dot(x1,x1)*0.25, dot(x2,x2)*0.25, max |x2-x1|). */
void orc_sor_stats(const float* x1, const float* x2, int n_power, float* stats)
{
    const int vh = vh_of(n_power);
    for (int t = 0; t < vh * vh; ++t) {
        const float* a = x1 + 4 * t;
        const float* b = x2 + 4 * t;
        stats[4 * t + 0] = (((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]) * 0.25f;
        stats[4 * t + 1] = (((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]) + a[3] * a[3]) * 0.25f;
        stats[4 * t + 2] = (((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]) + b[3] * b[3]) * 0.25f;
        float d0 = fabsf(b[0] - a[0]), d1 = fabsf(b[1] - a[1]), d2 = fabsf(b[2] - a[2]), d3 = fabsf(b[3] - a[3]);
        float m = d0 < d1 ? d1 : d0;
        m = m < d2 ? d2 : m;
        m = m < d3 ? d3 : m;
        stats[4 * t + 3] = m;
    }
}

/* out.solve (matrix_webgl.js:566-697).  R and C are rebuilt from A and b on every call; the
 * running sums x1, x2, x1x2, x1x1, x2x2 are NOT reset between iterations (kept); without
 * max_iterations (pass has_max = 0) `iteration < undefined` is false and nothing runs.
 * x_result is the state left by init_vector or an earlier solve; x_guess and stats are
 * outputs.  result[3] = { correlation, diff, iterations }. */
void orc_sor_solve(const float* A, const float* b, int n_power, double omega, double tolerance, int substep,
                   int has_max, int max_iterations, float* x_guess, float* x_result, float* x_stats, double result[3])
{
    const int vh = vh_of(n_power), mh = 2 * vh * vh, L = 4 * vh * vh, n_vec = vh * vh;
    float* R = (float*)malloc(sizeof(float) * 4 * (size_t)mh * mh);
    float* C = (float*)malloc(sizeof(float) * (size_t)L);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)L);
    orc_sor_build_R(A, n_power, omega, R);
    orc_sor_build_C(A, b, n_power, omega, C);
    double correlation = 0.0, x1 = 0, x2 = 0, x1x2 = 0, x1x1 = 0, x2x2 = 0;
    double diff = tolerance + 1;
    int iteration = 0;
    if (substep <= 0) substep = 1; /* params.substep || 1 */
    while (has_max && iteration < max_iterations && diff > tolerance) {
        for (int sub = 0; sub < substep; ++sub) {
            memcpy(x_guess, x_result, sizeof(float) * (size_t)L); /* programSet: x_result -> x_guess */
            orc_sor_mv_product(R, C, x_guess, n_power, omega, tmp);
            memcpy(x_result, tmp, sizeof(float) * (size_t)L);
        }
        orc_sor_stats(x_guess, x_result, n_power, x_stats);
        double max_diff = 0.0;
        for (int i = 0; i < n_vec; ++i) {
            x1 += (((double)x_guess[4 * i] + x_guess[4 * i + 1]) + x_guess[4 * i + 2]) + x_guess[4 * i + 3];
            x2 += (((double)x_result[4 * i] + x_result[4 * i + 1]) + x_result[4 * i + 2]) + x_result[4 * i + 3];
            x1x2 += x_stats[4 * i];
            x1x1 += x_stats[4 * i + 1];
            x2x2 += x_stats[4 * i + 2];
            max_diff = js_max(max_diff, (double)x_stats[4 * i + 3]);
        }
        correlation = (L * x1x2 - x1 * x2) / sqrt((L * x1x1 - x1 * x1) * (L * x2x2 - x2 * x2));
        diff = 2 * L * max_diff / (fabs(x1) + fabs(x2));
        iteration++;
    }
    result[0] = correlation;
    result[1] = diff;
    result[2] = iteration;
    free(R);
    free(C);
    free(tmp);
}
