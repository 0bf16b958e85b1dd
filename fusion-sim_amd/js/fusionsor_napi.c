/*
 * fusionsor_napi.c — N-API bindings of include/fusionsor.h (the dense iterative solver),
 * registered by fusionpic_napi.c under the names sor*.  The JavaScript shim matrix_native.js
 * turns them into the object the reference's makeSORIterative(spec) returns
 * (matrix_webgl.js:35-711).  A non-zero return of the C ABI becomes a synchronous JS Error.
 */
#include <node_api.h>
#include <stdlib.h>
#include <string.h>

#include "fusionsor.h"

typedef struct { fsor_handle* h; } sor_box_t;

static void sor_finalize(napi_env env, void* data, void* hint)
{
    (void)env; (void)hint;
    sor_box_t* b = (sor_box_t*)data;
    if (b) { if (b->h) fsor_destroy(b->h); free(b); }
}

static napi_value sor_throw(napi_env env, fsor_handle* h)
{
    const char* msg = fsor_last_error(h);
    napi_throw_error(env, NULL, (msg && *msg) ? msg : "libfusionpic solver call failed");
    return NULL;
}

static int sor_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv, fsor_handle** h)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    if (h) {
        sor_box_t* b = NULL;
        if (napi_get_value_external(env, argv[0], (void**)&b) != napi_ok || !b || !b->h) {
            napi_throw_error(env, NULL, "solver has been destroyed");
            return 0;
        }
        *h = b->h;
    }
    return 1;
}

static int sor_number(napi_env env, napi_value v, double* out)
{
    if (napi_get_value_double(env, v, out) != napi_ok) { napi_throw_type_error(env, NULL, "expected a number"); return 0; }
    return 1;
}

/* Float32Array or Float64Array of exactly `want` elements */
static int sor_floats(napi_env env, napi_value v, size_t want, void** data, int* dtype)
{
    bool is_ta = false;
    napi_typedarray_type t; size_t n, off; napi_value ab;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
        napi_get_typedarray_info(env, v, &t, &n, data, &ab, &off) != napi_ok ||
        (t != napi_float32_array && t != napi_float64_array)) {
        napi_throw_type_error(env, NULL, "expected Float32Array or Float64Array");
        return 0;
    }
    if (n != want) { napi_throw_range_error(env, NULL, "array length does not match the solver's size"); return 0; }
    *dtype = t == napi_float32_array ? FPIC_F32 : FPIC_F64;
    return 1;
}

static napi_value sor_undefined(napi_env env) { napi_value u; napi_get_undefined(env, &u); return u; }

/* sorCreate(n_power, relaxation (0 = not given), device, natural_rows) */
static napi_value n_sor_create(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; double d[4];
    if (!sor_args(env, info, 4, argv, NULL)) return NULL;
    for (int i = 0; i < 4; ++i) if (!sor_number(env, argv[i], &d[i])) return NULL;
    fsor_spec s; memset(&s, 0, sizeof s);
    s.n_power = (int32_t)d[0]; s.relaxation = d[1]; s.device = (int32_t)d[2]; s.natural_rows = (int32_t)d[3];
    fsor_handle* h = NULL;
    if (fsor_create(&s, &h) != FPIC_OK) return sor_throw(env, NULL);
    sor_box_t* b = (sor_box_t*)malloc(sizeof *b);
    if (!b) { fsor_destroy(h); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    b->h = h;
    napi_value ext;
    if (napi_create_external(env, b, sor_finalize, NULL, &ext) != napi_ok) { sor_finalize(env, b, NULL); napi_throw_error(env, NULL, "napi_create_external failed"); return NULL; }
    return ext;
}

static napi_value n_sor_destroy(napi_env env, napi_callback_info info)
{
    napi_value argv[1]; size_t argc = 1; sor_box_t* b = NULL;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) == napi_ok && argc == 1 &&
        napi_get_value_external(env, argv[0], (void**)&b) == napi_ok && b && b->h) { fsor_destroy(b->h); b->h = NULL; }
    return sor_undefined(env);
}

/* sorDims(h) -> [vec_length, vec_height] */
static napi_value n_sor_dims(napi_env env, napi_callback_info info)
{
    napi_value argv[1]; fsor_handle* h; uint64_t L; uint32_t vh;
    if (!sor_args(env, info, 1, argv, &h)) return NULL;
    if (fsor_dims(h, &L, &vh) != FPIC_OK) return sor_throw(env, h);
    napi_value arr, a, b;
    napi_create_array_with_length(env, 2, &arr);
    napi_create_double(env, (double)L, &a); napi_create_double(env, (double)vh, &b);
    napi_set_element(env, arr, 0, a); napi_set_element(env, arr, 1, b);
    return arr;
}

/* sorSet(h, what 0=matrix 1=b 2=x, typed array) */
static napi_value n_sor_set(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fsor_handle* h; double what; uint64_t L; void* data; int dtype;
    if (!sor_args(env, info, 3, argv, &h) || !sor_number(env, argv[1], &what)) return NULL;
    if (fsor_dims(h, &L, NULL) != FPIC_OK) return sor_throw(env, h);
    const size_t want = (int)what == 0 ? (size_t)(L * L) : (size_t)L;
    if (!sor_floats(env, argv[2], want, &data, &dtype)) return NULL;
    int rc = (int)what == 0 ? fsor_set_matrix(h, data, dtype) : (int)what == 1 ? fsor_set_b(h, data, dtype) : fsor_init_vector(h, data, dtype);
    if (rc != FPIC_OK) return sor_throw(env, h);
    return sor_undefined(env);
}

/* sorIterate(h, n) */
static napi_value n_sor_iterate(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; fsor_handle* h; double n;
    if (!sor_args(env, info, 2, argv, &h) || !sor_number(env, argv[1], &n)) return NULL;
    if (fsor_prepare(h) != FPIC_OK || fsor_iterate(h, (int32_t)n) != FPIC_OK) return sor_throw(env, h);
    return sor_undefined(env);
}

/* sorSolve(h, tolerance, substep (0 = not given), has_max, max_iterations, result Float32Array)
 * -> [correlation, diff, iterations] */
static napi_value n_sor_solve(napi_env env, napi_callback_info info)
{
    napi_value argv[6]; fsor_handle* h; double d[4]; uint64_t L; void* data; int dtype;
    if (!sor_args(env, info, 6, argv, &h)) return NULL;
    for (int i = 0; i < 4; ++i) if (!sor_number(env, argv[i + 1], &d[i])) return NULL;
    if (fsor_dims(h, &L, NULL) != FPIC_OK) return sor_throw(env, h);
    if (!sor_floats(env, argv[5], (size_t)L, &data, &dtype)) return NULL;
    if (dtype != FPIC_F32) { napi_throw_type_error(env, NULL, "result must be a Float32Array"); return NULL; }
    fsor_result r;
    if (fsor_solve(h, d[0], (int32_t)d[1], (int32_t)d[2], (int32_t)d[3], &r, (float*)data) != FPIC_OK) return sor_throw(env, h);
    napi_value arr, v;
    napi_create_array_with_length(env, 3, &arr);
    napi_create_double(env, r.correlation, &v); napi_set_element(env, arr, 0, v);
    napi_create_double(env, r.diff, &v); napi_set_element(env, arr, 1, v);
    napi_create_double(env, (double)r.iterations, &v); napi_set_element(env, arr, 2, v);
    return arr;
}

/* sorRead(h, which, out Float32Array) ; which = -1: the iteration matrix in texture layout */
static napi_value n_sor_read(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fsor_handle* h; double which; uint64_t L; void* data; int dtype;
    if (!sor_args(env, info, 3, argv, &h) || !sor_number(env, argv[1], &which)) return NULL;
    if (fsor_dims(h, &L, NULL) != FPIC_OK) return sor_throw(env, h);
    if (!sor_floats(env, argv[2], (int)which < 0 ? (size_t)(L * L) : (size_t)L, &data, &dtype)) return NULL;
    if (dtype != FPIC_F32) { napi_throw_type_error(env, NULL, "out must be a Float32Array"); return NULL; }
    int rc = (int)which < 0 ? fsor_read_iteration_matrix(h, (float*)data) : fsor_read_vector(h, (int)which, (float*)data);
    if (rc != FPIC_OK) return sor_throw(env, h);
    return sor_undefined(env);
}

static napi_value n_sor_sync(napi_env env, napi_callback_info info)
{
    napi_value argv[1]; fsor_handle* h;
    if (!sor_args(env, info, 1, argv, &h)) return NULL;
    if (fsor_sync(h) != FPIC_OK) return sor_throw(env, h);
    return sor_undefined(env);
}

int fusionsor_register(napi_env env, napi_value exports)
{
    struct { const char* name; napi_callback fn; } table[] = {
        { "sorCreate", n_sor_create }, { "sorDestroy", n_sor_destroy }, { "sorDims", n_sor_dims }, { "sorSet", n_sor_set },
        { "sorIterate", n_sor_iterate }, { "sorSolve", n_sor_solve }, { "sorRead", n_sor_read }, { "sorSync", n_sor_sync },
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i) {
        napi_value fn;
        if (napi_create_function(env, table[i].name, NAPI_AUTO_LENGTH, table[i].fn, NULL, &fn) != napi_ok ||
            napi_set_named_property(env, exports, table[i].name, fn) != napi_ok)
            return 0;
    }
    return 1;
}
