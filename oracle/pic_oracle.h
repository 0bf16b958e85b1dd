/*
 * pic_oracle.h — CPU oracle for fusion-sim's particle-in-cell hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a scalar C restatement of
 * the reference's GLSL programs and host routines (the reference has no CPU engine:
 * all arithmetic runs as WebGL shaders).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as
 * the checker.  libfusionpic.so never links, loads or calls anything in oracle/.
 *
 * PARITY PINNING.  The reference ships no tests, golden vectors or fixtures.  What pins this
 * restatement, strongest first:
 *   1. THE REFERENCE ITSELF RUN BY A REAL WebGL (round 4): oracle/make_golden_webgl.py executes the unmodified
 *      utilities.js / spindle.js / empic.js / matrix_webgl.js in the headless Chromium of the `kaleido` package (WebGL 1 on
 *      ANGLE/SwiftShader) and writes every frame buffer, read back with the reference's own readPixels, to
 *      tests/golden/webgl_*.  This restatement reproduces them bit for bit — upload, inverse CDF, precalc, every particle
 *      texel / random state / alive flag of every frame, the dense solver; the deposit under the rasterised convention
 *      (deposit_raster) — and within stated tolerances where the GL implementation's cos() and division enter
 *      (tests/test_oracle_webgl.py, tests/test_oracle_sor.py).
 *   2. the reference's own host JavaScript imported under Node (oracle/make_golden.js -> tests/golden/): the 11x11 stamp,
 *      the constants h / dt*c / factor_r / factor_z and the shader literals derived from them, the normalised particle
 *      upload, the E/B/sink texture packing, the 512x512 inverse-CDF table (NaN sites included), and the pass order and
 *      per-pass texture bindings of step() and density();
 *   3. the reference's shader text under a software evaluator (oracle/swgl.js, oracle/glsl_eval.js; IEEE float32 per
 *      operation): tests/golden/swgl_*, reproduced bit for bit (tests/test_oracle_swgl.py) — a transcription check, itself
 *      checked against 1. since round 4.
 * What stays implementation-defined in GLSL ES 1.00 and therefore differs between GL implementations: transcendental
 * built-ins, division by a varying, and the rasteriser's sub-pixel grid.
 *
 * Two instantiations of every kernel: orc_f32_* (float, the reference's precision)
 * and orc_f64_* (double).  Arrays are RGBA textures, texel (i,j) at 4*(i + j*W).
 */
#ifndef PIC_ORACLE_H
#define PIC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_N_ENTROPY 1024 /* empic.js:142 */
#define ORC_N_CDF 512      /* empic.js:228-241 */
#define ORC_NSHAPE 11      /* empic.js:949 */

/* Constants of the factory (empic.js:27, :44-46, :852):
 * out = { h, factor_r, factor_z, dt*c, factor_r/factor_z, factor_z/factor_r } in double. */
void orc_constants(double radius, double height, double dt, double mass, double charge, double out[6]);

/* The literal N(x) = x.toFixed(20) as a GLSL float would read it (empic.js:23-25). */
double orc_tofixed20(double x);

/* 11x11 cos^2 stamp, red channel, 121 floats, index i + 11*j (empic.js:949-971). */
void orc_stamp(float out[121]);

/* Inverse-CDF table of out.set({source_pdf}) (empic.js:1263-1339).  pdf is
 * value[i][j] flattened (i over nr, j over nz); out is 512*512*4 floats with x,y in
 * channels 0,1 (channels 2,3 stay 0).  Returns 0, or -1 if the reference would
 * throw "function out of range". */
int orc_inv_cdf(const double* pdf, int nr, int nz, float* out);

/* Philox4x32-10 block: counter (c0..c3), key (k0,k1) -> 4 words (RNG extension mode). */
void orc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);

#define ORC_DECL(REAL, P)                                                                              \
    void P##step_rand(const REAL* rand_in, const REAL* entropy, REAL* rand_out, size_t n);             \
    void P##step_velocity(const REAL* pos, const REAL* vel, const REAL* rnd, const REAL* R1,           \
                          const REAL* R2, const REAL* R3, const REAL* A, int nr, int nz,               \
                          REAL* vel_out, size_t n);                                                    \
    void P##step_position(const REAL* pos, const REAL* vel_new, const REAL* rnd, const REAL* sink,     \
                          const REAL* inv_cdf, int nr, int nz, REAL step_factor, REAL* pos_out,        \
                          size_t n);                                                                   \
    void P##step(REAL* pos_A, REAL* vel_A, REAL* rand_A, REAL* pos_B, REAL* vel_B, REAL* rand_B,       \
                 const REAL* entropy, const REAL* R1, const REAL* R2, const REAL* R3, const REAL* A,   \
                 const REAL* sink, const REAL* inv_cdf, int nr, int nz, REAL step_factor, size_t n,    \
                 int ncalls);                                                                          \
    void P##cells(const REAL* pos, size_t n, int nr, int nz, int32_t* cells);                          \
    void P##precalc(const REAL* B, const REAL* E, int nr, int nz, REAL h, REAL factor_r,               \
                    REAL factor_z, REAL f_rz, REAL f_zr, REAL* R1, REAL* R2, REAL* R3, REAL* A,        \
                    int physical_a);                                                                   \
    void P##deposit(const REAL* pos, const REAL* vel, size_t n, const float* stamp, int nr, int nz,    \
                    REAL* moments);                                                                    \
    void P##deposit_raster(const REAL* pos, const REAL* vel, size_t n, const float* stamp, int nr,     \
                           int nz, REAL* moments, int subpixel_bits);                                 \
    void P##raster_cells(const REAL* pos, size_t n, int nr, int nz, int subpixel_bits, int32_t* ci,    \
                         int32_t* cj);                                                                 \
    void P##deposit_cic(const REAL* pos, const REAL* vel, size_t n, int nr, int nz, REAL* moments);    \
    void P##deposit_cells(const REAL* pos, size_t n, int nr, int nz, int32_t* cells);                  \
    void P##normalise(const REAL* moments, int nr, int nz, REAL* norm);                                \
    void P##avg(const REAL* next, REAL* avg_B, REAL* avg_A, REAL ratio, size_t ncell);                 \
    void P##normalise_particles(const double* aos3, size_t n, double factor_xy, double factor_z,       \
                                REAL* out4);                                                           \
    void P##pack_grid3(const double* in_ijk, int nr, int nz, REAL* rgba);                              \
    void P##pack_sink(const double* in_ij, int nr, int nz, REAL* rgba);                                \
    void P##loop_shape(REAL u_R, int nr, int nz, REAL* out);                                           \
    void P##add_current_loop(REAL* B, const REAL* shape_half, const REAL* shape_tenth, int nr, int nz, \
                             REAL u_R, REAL u_Z, REAL u_I);                                            \
    void P##add_uniform(REAL* B, int nr, int nz, int kind, REAL value);                                \
    void P##step_counter(REAL* pos_A, REAL* vel_A, REAL* pos_B, REAL* vel_B, const REAL* R1,           \
                         const REAL* R2, const REAL* R3, const REAL* A, const REAL* sink,              \
                         const REAL* inv_cdf, int nr, int nz, REAL step_factor, size_t n, int ncalls,  \
                         uint64_t seed, uint64_t t0);

ORC_DECL(float, orc_f32_)
ORC_DECL(double, orc_f64_)

#ifdef __cplusplus
}
#endif
#endif
