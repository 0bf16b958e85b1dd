/*
 * fusionsor.h — C ABI of the dense iterative solver in libfusionpic.so (MI355X / gfx950).
 *
 * Drop-in boundary for SURVEY.md 8(f) next-4: the reference's
 *     matrix_webgl.makeSORIterative(spec)          (matrix_webgl.js:35-711)
 * a weighted-Jacobi iteration x <- omega (R x + C) + (1 - omega) x on a dense
 * L x L system, L = 4 * (2^n_power)^2, all arithmetic float32.  The functions below are what a
 * JavaScript (N-API), Python (ctypes) or C host binds instead of the reference's WebGL
 * programs; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions are those of fusionpic.h: every call returns FPIC_OK (0) or a negative
 * fpic_status, the message is available from fsor_last_error(); no exception crosses the
 * boundary; host buffers are caller-owned and copied during the call; a handle is not
 * thread-safe; there is no CPU fallback.
 *
 * Bit-exactness: the per-row sums are formed in the order the reference's shader passes form
 * them (a radix-4 tree over the row's vh x vh texel block, per colour channel, then
 * ((r+g)+b)+a), so results equal the restatement in oracle/sor_oracle.c bit for bit.
 */
#ifndef FUSIONSOR_H
#define FUSIONSOR_H

#include "fusionpic.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FSOR_ABI_VERSION 1

typedef struct fsor_handle fsor_handle;

/* spec of makeSORIterative (matrix_webgl.js:36-40) + extensions */
typedef struct fsor_spec {
    int32_t n_power;      /* length(vector) = 4 * (2^n_power)^2; 1..7 (0 makes the reference throw) */
    int32_t device;       /* HIP device ordinal */
    double relaxation;    /* omega; 0 means "not given" = 1.0 (`spec.relaxation || 1.0`, :54) */
    int32_t natural_rows; /* 0 (default): keep the reference's row permutation of the update
                             (quirk Q14, matrix_webgl.js:389-424); 1: element e uses matrix row e,
                             i.e. the iteration actually converges to the solution of A x = b */
    int32_t reserved[5];
} fsor_spec;

/* what out.solve() returns (matrix_webgl.js:689-694); `result` is copied separately */
typedef struct fsor_result {
    double correlation;
    double diff;
    int32_t iterations;
    int32_t reserved;
} fsor_result;

typedef enum fsor_vector {
    FSOR_X_RESULT = 0, /* x_result: the current iterate (matrix_webgl.js:140) */
    FSOR_X_GUESS = 1,  /* x_guess: the previous iterate (:134) */
    FSOR_X_STATS = 2,  /* x_stats: per texel (x1.x2/4, x1.x1/4, x2.x2/4, max|x2-x1|) (:146, :428-452) */
    FSOR_C = 3,        /* the constant vector C (:173, :266-301); valid after fsor_prepare/solve */
    FSOR_B = 4         /* b as uploaded */
} fsor_vector;

typedef struct fsor_stats {
    uint64_t iterations;    /* matrix-vector products since creation / reset */
    double seconds_iterate; /* device time of those launches; filled while profiling is on */
    uint64_t matrix_bytes;  /* bytes of the iteration matrix one product streams */
} fsor_stats;

const char* fsor_last_error(const fsor_handle* h); /* h may be NULL: last create error of this thread */
int fsor_abi_version(void);

/* makeSORIterative(spec) (matrix_webgl.js:35-455): allocates A, R (L*L floats each) and the vectors */
int fsor_create(const fsor_spec* spec, fsor_handle** out);
void fsor_destroy(fsor_handle* h);
/* out.vec_length / out.vec_height (matrix_webgl.js:50-51) */
int fsor_dims(const fsor_handle* h, uint64_t* vec_length, uint32_t* vec_height);

/* out.set_matrix(matrix) (:456-475): row-major A[col + L*row], dtype FPIC_F32 or FPIC_F64 (rounded to float) */
int fsor_set_matrix(fsor_handle* h, const void* a_row_major, int dtype);
/* out.set_b(b) (:479-496) */
int fsor_set_b(fsor_handle* h, const void* b, int dtype);
/* out.init_vector(vector) (:500-530): x_result <- vector */
int fsor_init_vector(fsor_handle* h, const void* x, int dtype);

/* the first half of out.solve(): R <- programR(A), C <- programC(A, b) (:573-588) */
int fsor_prepare(fsor_handle* h);
/* n times { x_guess <- x_result; x_result <- out.mv_product(x_result) } (:629-639) without statistics
 * or read-back; asynchronous on the handle's stream.  Requires fsor_prepare. */
int fsor_iterate(fsor_handle* h, int32_t n);
/* out.solve({tolerance, substep, max_iterations}) (:566-697).  substep <= 0 means "not given" (1);
 * has_max_iterations = 0 reproduces the reference's behaviour without it (no iteration runs).
 * result (may be NULL) receives vec_length floats: the closure array x2_arr. */
int fsor_solve(fsor_handle* h, double tolerance, int32_t substep, int32_t has_max_iterations, int32_t max_iterations,
               fsor_result* out, float* result);

/* read-back (the reference only has readPixels on its frame buffers, utilities.js:701-711) */
int fsor_read_vector(fsor_handle* h, int which, float* out);
/* the iteration matrix in the reference's texture layout, 4*(2 vh^2)^2 floats (:165-169, :222-262) */
int fsor_read_iteration_matrix(fsor_handle* h, float* out);
/* out.x_result_tex() (:701-704): the device allocation behind x_result, for device-side consumers */
int fsor_device_buffer(fsor_handle* h, int which, void** ptr, size_t* bytes);

int fsor_set_stream(fsor_handle* h, void* hip_stream);
int fsor_sync(fsor_handle* h);
int fsor_profile(fsor_handle* h, int enable);
int fsor_get_stats(fsor_handle* h, fsor_stats* out);
int fsor_reset_stats(fsor_handle* h);

#ifdef __cplusplus
}
#endif

#endif /* FUSIONSOR_H */
