/*
 * fusionpic_napi.c — Node N-API (version 8) addon over the C ABI of libfusionpic.so.
 *
 * The reference's host language is JavaScript; its hot path is the object returned
 * by empic.makeCylindricalParticlePusher (empic.js:30-1529).  This addon is the thin
 * native layer under fusion-sim_amd/js/empic_native.js, which re-creates that object
 * with the same method names.  Every function here is one ABI call: it unpacks
 * numbers and typed arrays, calls fpic_*, and converts a non-zero status into a
 * synchronous JavaScript Error carrying fpic_last_error() — the way the reference
 * reports validation and GL failures (utilities.js:118-127, :213-259).
 *
 * Ownership: typed arrays stay owned by JavaScript and are only read or filled
 * during the call.  The native handle is wrapped in an external whose finaliser
 * calls fpic_destroy (napi_add_finalizer semantics), so an unreachable pusher frees
 * its HBM; destroy() does it eagerly.
 */
#include <node_api.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "fusionpic.h"

#define NAPI_OK(env, call)                                                        \
    do {                                                                          \
        if ((call) != napi_ok) {                                                  \
            napi_throw_error((env), NULL, "N-API call failed: " #call);           \
            return NULL;                                                          \
        }                                                                         \
    } while (0)

#define MAX_SPECIES 16
typedef struct {
    fpic_handle* h;
    size_t n;     /* particle count of the handle: every typed array is checked against the */
    size_t cells; /* size the ABI call will read or write before its pointer is passed on   */
    /* CART3D extension: node count and the particle count of every species */
    size_t nodes;
    int nspecies;
    size_t count[MAX_SPECIES];
} box_t;

static void box_finalize(napi_env env, void* data, void* hint)
{
    (void)env; (void)hint;
    box_t* b = (box_t*)data;
    if (b) {
        if (b->h) fpic_destroy(b->h);
        free(b);
    }
}

static napi_value throw_fpic(napi_env env, fpic_handle* h)
{
    const char* msg = fpic_last_error(h);
    napi_throw_error(env, NULL, (msg && *msg) ? msg : "libfusionpic call failed");
    return NULL;
}

static __thread box_t* g_box; /* box of the call being served on this JS thread (set by get_args) */

/* A typed array whose length differs from what the ABI call touches is a RangeError in
 * JavaScript, never a native out-of-bounds access. */
static int check_len(napi_env env, const char* name, size_t len, size_t want)
{
    if (len == want) return 1;
    char msg[160];
    snprintf(msg, sizeof msg, ".%s <- expected a typed array of %zu elements, got %zu", name, want, len);
    napi_throw_range_error(env, NULL, msg);
    return 0;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv, fpic_handle** h)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    if (h) {
        box_t* b = NULL;
        if (napi_get_value_external(env, argv[0], (void**)&b) != napi_ok || !b || !b->h) {
            napi_throw_error(env, NULL, "pusher has been destroyed");
            return 0;
        }
        *h = b->h;
        g_box = b;
    }
    return 1;
}

static int get_double(napi_env env, napi_value v, double* out)
{
    if (napi_get_value_double(env, v, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "expected a number");
        return 0;
    }
    return 1;
}

/* Float32Array / Float64Array / Uint8Array / Int32Array or null/undefined. */
static int get_typed(napi_env env, napi_value v, napi_typedarray_type* type, void** data, size_t* length)
{
    napi_valuetype vt;
    *data = NULL; *length = 0;
    if (napi_typeof(env, v, &vt) != napi_ok) return 0;
    if (vt == napi_undefined || vt == napi_null) return 1;
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) {
        napi_throw_type_error(env, NULL, "expected a typed array");
        return 0;
    }
    napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, v, type, length, data, &ab, &off) != napi_ok) {
        napi_throw_type_error(env, NULL, "bad typed array");
        return 0;
    }
    return 1;
}

static int float_dtype(napi_env env, napi_typedarray_type t, int* dtype)
{
    if (t == napi_float32_array) { *dtype = FPIC_F32; return 1; }
    if (t == napi_float64_array) { *dtype = FPIC_F64; return 1; }
    napi_throw_type_error(env, NULL, "expected Float32Array or Float64Array");
    return 0;
}

static napi_value undefined(napi_env env)
{
    napi_value u;
    napi_get_undefined(env, &u);
    return u;
}

/* create(radius, height, nr, nz, dt, nparticles, mass, charge, count, precision, device, physical_a,
 *        sort_interval, unfused_deposit, rng_mode, seed_lo, seed_hi, geometry, solver, ny, length_y, macro_weight, shape,
 *        raster_subpixel_bits) */
static napi_value n_create(napi_env env, napi_callback_info info)
{
    napi_value argv[24];
    if (!get_args(env, info, 24, argv, NULL)) return NULL;
    double d[24];
    for (int i = 0; i < 24; ++i) if (!get_double(env, argv[i], &d[i])) return NULL;
    fpic_spec s;
    memset(&s, 0, sizeof s);
    s.radius = d[0]; s.height = d[1]; s.nr = (int32_t)d[2]; s.nz = (int32_t)d[3]; s.dt = d[4];
    s.nparticles = (int32_t)d[5]; s.particle_mass = d[6]; s.particle_charge = d[7];
    s.count = (uint64_t)d[8]; s.precision = (int32_t)d[9]; s.device = (int32_t)d[10];
    s.physical_a = (int32_t)d[11]; s.sort_interval = (int32_t)d[12]; s.unfused_deposit = (int32_t)d[13];
    s.rng_mode = (int32_t)d[14]; s.rng_seed_lo = (uint32_t)d[15]; s.rng_seed_hi = (uint32_t)d[16];
    s.geometry = (int32_t)d[17]; s.solver = (int32_t)d[18]; s.ny = (int32_t)d[19]; s.length_y = d[20]; s.macro_weight = d[21]; s.shape = (int32_t)d[22];
    s.raster_subpixel_bits = (int32_t)d[23];
    fpic_handle* h = NULL;
    if (fpic_create(&s, &h) != FPIC_OK) return throw_fpic(env, NULL);
    box_t* b = (box_t*)malloc(sizeof *b);
    if (!b) { fpic_destroy(h); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    b->h = h;
    b->n = s.count ? (size_t)s.count : (size_t)s.nparticles * (size_t)s.nparticles;
    b->cells = (size_t)s.nr * (size_t)s.nz;
    b->nodes = b->cells * (size_t)(s.ny > 0 ? s.ny : 1);
    b->nspecies = 1;
    b->count[0] = b->n;
    napi_value ext;
    if (napi_create_external(env, b, box_finalize, NULL, &ext) != napi_ok) {
        box_finalize(env, b, NULL);
        napi_throw_error(env, NULL, "napi_create_external failed");
        return NULL;
    }
    return ext;
}

static napi_value n_destroy(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    size_t argc = 1;
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    box_t* b = NULL;
    if (argc == 1 && napi_get_value_external(env, argv[0], (void**)&b) == napi_ok && b && b->h) {
        fpic_destroy(b->h);
        b->h = NULL;
    }
    return undefined(env);
}

/* setParticles(h, positionAoS|null, velocityAoS|null) */
static napi_value n_set_particles(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fpic_handle* h;
    if (!get_args(env, info, 3, argv, &h)) return NULL;
    for (int k = 0; k < 2; ++k) {
        napi_typedarray_type t; void* data; size_t len;
        if (!get_typed(env, argv[1 + k], &t, &data, &len)) return NULL;
        if (!data) continue;
        int dtype;
        if (!float_dtype(env, t, &dtype)) return NULL;
        if (len % 3) { napi_throw_error(env, NULL, ".position <- length must be a multiple of 3"); return NULL; }
        if (fpic_set_particles(h, k == 0 ? data : NULL, k == 1 ? data : NULL, len / 3, dtype) != FPIC_OK) return throw_fpic(env, h);
    }
    return undefined(env);
}

/* setGrid(h, which, data, nr, nz, ncomp) */
static napi_value n_set_grid(napi_env env, napi_callback_info info)
{
    napi_value argv[6]; fpic_handle* h;
    if (!get_args(env, info, 6, argv, &h)) return NULL;
    double which, nr, nz, nc;
    if (!get_double(env, argv[1], &which) || !get_double(env, argv[3], &nr) || !get_double(env, argv[4], &nz) ||
        !get_double(env, argv[5], &nc)) return NULL;
    napi_typedarray_type t; void* data; size_t len; int dtype;
    if (!get_typed(env, argv[2], &t, &data, &len)) return NULL;
    if (!data) { napi_throw_type_error(env, NULL, "expected a typed array"); return NULL; }
    if (!float_dtype(env, t, &dtype)) return NULL;
    if (len != (size_t)(nr * nz * nc)) { napi_throw_error(env, NULL, ".grid <- wrong number of elements"); return NULL; }
    if (fpic_set_grid(h, (int)which, data, (int)nr, (int)nz, (int)nc, dtype) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* setRandomState(h, entropy Float32Array|null, rand Float32Array|null) */
static napi_value n_set_random_state(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fpic_handle* h;
    if (!get_args(env, info, 3, argv, &h)) return NULL;
    void* ptr[2] = { NULL, NULL };
    for (int k = 0; k < 2; ++k) {
        napi_typedarray_type t; size_t len;
        if (!get_typed(env, argv[1 + k], &t, &ptr[k], &len)) return NULL;
        if (ptr[k] && t != napi_float32_array) { napi_throw_type_error(env, NULL, "expected Float32Array"); return NULL; }
        if (ptr[k] && k == 0 && len != (size_t)4 * 1024 * 1024) { napi_throw_error(env, NULL, ".entropy <- expected 1024*1024*4 floats"); return NULL; }
        if (ptr[k] && k == 1 && !check_len(env, "rand", len, 4 * g_box->n)) return NULL;
    }
    if (fpic_set_random_state(h, (const float*)ptr[0], (const float*)ptr[1]) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

#define SIMPLE_CALL(NAME, FN)                                                \
    static napi_value NAME(napi_env env, napi_callback_info info)            \
    {                                                                        \
        napi_value argv[1]; fpic_handle* h;                                  \
        if (!get_args(env, info, 1, argv, &h)) return NULL;                  \
        if (FN(h) != FPIC_OK) return throw_fpic(env, h);                     \
        return undefined(env);                                               \
    }
SIMPLE_CALL(n_precalc, fpic_precalc)
SIMPLE_CALL(n_density, fpic_density)
SIMPLE_CALL(n_deposit, fpic_deposit)
SIMPLE_CALL(n_density_finish, fpic_density_finish)
SIMPLE_CALL(n_sort, fpic_sort)
SIMPLE_CALL(n_sync, fpic_sync)
SIMPLE_CALL(n_reset_stats, fpic_reset_stats)

#define DOUBLE_CALL(NAME, FN)                                                \
    static napi_value NAME(napi_env env, napi_callback_info info)            \
    {                                                                        \
        napi_value argv[2]; fpic_handle* h; double v;                        \
        if (!get_args(env, info, 2, argv, &h)) return NULL;                  \
        if (!get_double(env, argv[1], &v)) return NULL;                      \
        if (FN(h, v) != FPIC_OK) return throw_fpic(env, h);                  \
        return undefined(env);                                               \
    }
DOUBLE_CALL(n_add_current_z, fpic_add_current_z)
DOUBLE_CALL(n_add_bz, fpic_add_bz)
DOUBLE_CALL(n_add_btheta, fpic_add_btheta)

static napi_value n_add_current_loop(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; fpic_handle* h; double r, z, c;
    if (!get_args(env, info, 4, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &r) || !get_double(env, argv[2], &z) || !get_double(env, argv[3], &c)) return NULL;
    if (fpic_add_current_loop(h, r, z, c) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

static napi_value n_step(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; fpic_handle* h; double n;
    if (!get_args(env, info, 2, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &n)) return NULL;
    if (fpic_step(h, (int)n) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

static napi_value n_substeps(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; fpic_handle* h; double n;
    if (!get_args(env, info, 2, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &n)) return NULL;
    if (fpic_substeps(h, (int)n) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* stepAsync(h, ncalls) -> Promise: fpic_step + fpic_sync on a libuv worker thread, so that Node's loop is not blocked
 * while the GPU works ("Threading").  The handle is not thread-safe: the shim refuses other calls on the
 * simulation until the promise has settled. */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref keep;   /* the handle's JS object stays alive (no finaliser) while the step runs */
    fpic_handle* h;
    int ncalls, rc;
    char message[512];
} step_job;

static void step_execute(napi_env env, void* data)
{
    (void)env;
    step_job* j = (step_job*)data;
    j->rc = fpic_step(j->h, j->ncalls);
    if (j->rc == FPIC_OK) j->rc = fpic_sync(j->h);
    if (j->rc != FPIC_OK) {
        const char* m = fpic_last_error(j->h);
        strncpy(j->message, m ? m : "fpic_step failed", sizeof j->message - 1);
        j->message[sizeof j->message - 1] = 0;
    }
}

static void step_complete(napi_env env, napi_status status, void* data)
{
    step_job* j = (step_job*)data;
    napi_value v;
    if (status == napi_ok && j->rc == FPIC_OK) {
        napi_get_undefined(env, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg;
        napi_create_string_utf8(env, status == napi_ok ? j->message : "the step was cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &v);
        napi_reject_deferred(env, j->deferred, v);
    }
    napi_delete_reference(env, j->keep);
    napi_delete_async_work(env, j->work);
    free(j);
}

static napi_value n_step_async(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; fpic_handle* h; double n;
    if (!get_args(env, info, 2, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &n)) return NULL;
    step_job* j = (step_job*)calloc(1, sizeof *j);
    if (!j) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    j->h = h; j->ncalls = (int)n;
    napi_value promise, name;
    if (napi_create_reference(env, argv[0], 1, &j->keep) != napi_ok) { free(j); napi_throw_error(env, NULL, "cannot reference the handle"); return NULL; }
    NAPI_OK(env, napi_create_promise(env, &j->deferred, &promise));
    NAPI_OK(env, napi_create_string_utf8(env, "fusionpic.step", NAPI_AUTO_LENGTH, &name));
    NAPI_OK(env, napi_create_async_work(env, NULL, name, step_execute, step_complete, j, &j->work));
    NAPI_OK(env, napi_queue_async_work(env, j->work));
    return promise;
}

static napi_value n_profile(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; fpic_handle* h; double on;
    if (!get_args(env, info, 2, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &on)) return NULL;
    if (fpic_profile(h, (int)on) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* readGrid(h, which, out Float32Array|Float64Array) */
static napi_value n_read_grid(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fpic_handle* h; double which;
    if (!get_args(env, info, 3, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &which)) return NULL;
    napi_typedarray_type t; void* data; size_t len; int dtype;
    if (!get_typed(env, argv[2], &t, &data, &len)) return NULL;
    if (!data) { napi_throw_type_error(env, NULL, "expected a typed array"); return NULL; }
    if (!float_dtype(env, t, &dtype)) return NULL;
    if (!check_len(env, "out", len, 4 * ((int)which == FPIC_READ_INV_CDF ? (size_t)512 * 512 : g_box->cells))) return NULL;
    if (fpic_read_grid(h, (int)which, data, dtype) != FPIC_OK) return throw_fpic(env, h);
    return argv[2];
}

/* getParticles(h, position|null, velocity|null, rand Float32Array|null, alive Uint8Array|null) */
static napi_value n_get_particles(napi_env env, napi_callback_info info)
{
    napi_value argv[5]; fpic_handle* h;
    if (!get_args(env, info, 5, argv, &h)) return NULL;
    napi_typedarray_type t[4]; void* p[4]; size_t len[4];
    for (int k = 0; k < 4; ++k) if (!get_typed(env, argv[1 + k], &t[k], &p[k], &len[k])) return NULL;
    int dtype = FPIC_F32;
    if (p[0] && !float_dtype(env, t[0], &dtype)) return NULL;
    if (p[1]) {
        int d2;
        if (!float_dtype(env, t[1], &d2)) return NULL;
        if (p[0] && d2 != dtype) { napi_throw_type_error(env, NULL, "position and velocity must have the same element type"); return NULL; }
        dtype = d2;
    }
    if (p[2] && t[2] != napi_float32_array) { napi_throw_type_error(env, NULL, "rand must be a Float32Array"); return NULL; }
    if (p[3] && t[3] != napi_uint8_array) { napi_throw_type_error(env, NULL, "alive must be a Uint8Array"); return NULL; }
    static const char* const names[4] = { "position", "velocity", "rand", "alive" };
    static const size_t per[4] = { 3, 3, 4, 1 };
    for (int k = 0; k < 4; ++k)
        if (p[k] && !check_len(env, names[k], len[k], per[k] * g_box->n)) return NULL;
    if (fpic_get_particles(h, p[0], p[1], (float*)p[2], (uint8_t*)p[3], dtype) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

static napi_value n_get_cells(napi_env env, napi_callback_info info)
{
    napi_value argv[2]; fpic_handle* h;
    if (!get_args(env, info, 2, argv, &h)) return NULL;
    napi_typedarray_type t; void* p; size_t len;
    if (!get_typed(env, argv[1], &t, &p, &len)) return NULL;
    if (!p || t != napi_int32_array) { napi_throw_type_error(env, NULL, "expected an Int32Array"); return NULL; }
    if (!check_len(env, "cells", len, g_box->n)) return NULL;
    if (fpic_get_cells(h, (int32_t*)p) != FPIC_OK) return throw_fpic(env, h);
    return argv[1];
}

/* saveCheckpoint(h, path) / loadCheckpoint(h, path) */
static napi_value checkpoint_call(napi_env env, napi_callback_info info, int (*fn)(fpic_handle*, const char*))
{
    napi_value argv[2]; fpic_handle* h;
    if (!get_args(env, info, 2, argv, &h)) return NULL;
    char path[4096]; size_t len = 0;
    if (napi_get_value_string_utf8(env, argv[1], path, sizeof path, &len) != napi_ok) {
        napi_throw_type_error(env, NULL, "expected a path string");
        return NULL;
    }
    if (fn(h, path) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}
static napi_value n_save_checkpoint(napi_env env, napi_callback_info info) { return checkpoint_call(env, info, fpic_save_checkpoint); }
static napi_value n_load_checkpoint(napi_env env, napi_callback_info info) { return checkpoint_call(env, info, fpic_load_checkpoint); }

static napi_value n_get_stats(napi_env env, napi_callback_info info)
{
    napi_value argv[1]; fpic_handle* h;
    if (!get_args(env, info, 1, argv, &h)) return NULL;
    fpic_stats s;
    if (fpic_get_stats(h, &s) != FPIC_OK) return throw_fpic(env, h);
    napi_value o;
    NAPI_OK(env, napi_create_object(env, &o));
    struct { const char* k; double v; } f[] = {
        { "n_particles", (double)s.n_particles }, { "particle_updates", (double)s.particle_updates },
        { "step_launches", (double)s.step_launches }, { "deposit_launches", (double)s.deposit_launches },
        { "sort_passes", (double)s.sort_passes }, { "deposit_spilled", (double)s.deposit_spilled },
        { "ms_push", s.ms_push }, { "ms_deposit", s.ms_deposit }, { "ms_stamp", s.ms_stamp },
        { "ms_precalc", s.ms_precalc }, { "ms_sort", s.ms_sort },
        { "bytes_particle_state", (double)s.bytes_particle_state }, { "bytes_grid_state", (double)s.bytes_grid_state },
    };
    for (size_t i = 0; i < sizeof f / sizeof f[0]; ++i) {
        napi_value v;
        NAPI_OK(env, napi_create_double(env, f[i].v, &v));
        NAPI_OK(env, napi_set_named_property(env, o, f[i].k, v));
    }
    return o;
}

/* ---- CART3D extension (include/fusionpic.h, "extension: spec.geometry") ---- */
static int get_species(napi_env env, napi_value v, int* sp)
{
    double d;
    if (!get_double(env, v, &d)) return 0;
    if (d < 0 || d >= g_box->nspecies) { napi_throw_range_error(env, NULL, ".species <- no such species"); return 0; }
    *sp = (int)d;
    return 1;
}

/* addSpecies(h, mass, charge, count) -> index */
static napi_value n_add_species(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; fpic_handle* h; double m, q, c;
    if (!get_args(env, info, 4, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &m) || !get_double(env, argv[2], &q) || !get_double(env, argv[3], &c)) return NULL;
    if (g_box->nspecies >= MAX_SPECIES) { napi_throw_range_error(env, NULL, ".species <- too many species"); return NULL; }
    int idx = 0;
    if (fpic_add_species(h, m, q, (uint64_t)c, &idx) != FPIC_OK) return throw_fpic(env, h);
    g_box->count[g_box->nspecies++] = (size_t)c;
    napi_value out;
    NAPI_OK(env, napi_create_int32(env, idx, &out));
    return out;
}

/* setParticlesRange(h, species, first, position|null, velocity|null): the caller's particles [first, first + len/3) */
static napi_value n_set_particles_range(napi_env env, napi_callback_info info)
{
    napi_value argv[5]; fpic_handle* h; int sp; double first;
    if (!get_args(env, info, 5, argv, &h)) return NULL;
    if (!get_species(env, argv[1], &sp) || !get_double(env, argv[2], &first)) return NULL;
    napi_typedarray_type t[2]; void* p[2]; size_t len[2]; int dtype = -1;
    for (int k = 0; k < 2; ++k) {
        if (!get_typed(env, argv[3 + k], &t[k], &p[k], &len[k])) return NULL;
        if (!p[k]) continue;
        int d2;
        if (!float_dtype(env, t[k], &d2)) return NULL;
        if (dtype >= 0 && d2 != dtype) { napi_throw_type_error(env, NULL, "position and velocity must have the same element type"); return NULL; }
        dtype = d2;
        if (len[k] % 3) { napi_throw_range_error(env, NULL, ".position <- length must be a multiple of 3"); return NULL; }
    }
    if (dtype < 0) return undefined(env);
    if (p[0] && p[1] && len[0] != len[1]) { napi_throw_range_error(env, NULL, ".velocity <- must be as long as .position"); return NULL; }
    const size_t m = (p[0] ? len[0] : len[1]) / 3;
    if (first < 0 || (size_t)first > g_box->count[sp] || m > g_box->count[sp] - (size_t)first) {
        napi_throw_range_error(env, NULL, ".position <- range lies outside the species");
        return NULL;
    }
    if (fpic_set_particles_range(h, sp, (uint64_t)first, m, p[0], p[1], dtype) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* getParticlesOf(h, species, position|null, velocity|null) */
static napi_value n_get_particles_of(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; fpic_handle* h; int sp;
    if (!get_args(env, info, 4, argv, &h)) return NULL;
    if (!get_species(env, argv[1], &sp)) return NULL;
    napi_typedarray_type t[2]; void* p[2]; size_t len[2]; int dtype = FPIC_F32, have = 0;
    for (int k = 0; k < 2; ++k) {
        if (!get_typed(env, argv[2 + k], &t[k], &p[k], &len[k])) return NULL;
        if (!p[k]) continue;
        int d2;
        if (!float_dtype(env, t[k], &d2)) return NULL;
        if (have && d2 != dtype) { napi_throw_type_error(env, NULL, "position and velocity must have the same element type"); return NULL; }
        dtype = d2; have = 1;
        if (!check_len(env, k == 0 ? "position" : "velocity", len[k], 3 * g_box->count[sp])) return NULL;
    }
    if (fpic_get_particles_of(h, sp, p[0], p[1], dtype) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* getParticlesRange(h, species, first, stride, position|null, velocity|null): the caller's particles first, first + stride, ...
 * — as many as the arrays hold (len / 3) */
static napi_value n_get_particles_range(napi_env env, napi_callback_info info)
{
    napi_value argv[6]; fpic_handle* h; int sp; double first, stride;
    if (!get_args(env, info, 6, argv, &h)) return NULL;
    if (!get_species(env, argv[1], &sp) || !get_double(env, argv[2], &first) || !get_double(env, argv[3], &stride)) return NULL;
    napi_typedarray_type t[2]; void* p[2]; size_t len[2] = { 0, 0 }; int dtype = -1;
    for (int k = 0; k < 2; ++k) {
        if (!get_typed(env, argv[4 + k], &t[k], &p[k], &len[k])) return NULL;
        if (!p[k]) continue;
        int d2;
        if (!float_dtype(env, t[k], &d2)) return NULL;
        if (dtype >= 0 && d2 != dtype) { napi_throw_type_error(env, NULL, "position and velocity must have the same element type"); return NULL; }
        dtype = d2;
        if (len[k] % 3) { napi_throw_range_error(env, NULL, ".position <- length must be a multiple of 3"); return NULL; }
    }
    if (dtype < 0) return undefined(env);
    if (p[0] && p[1] && len[0] != len[1]) { napi_throw_range_error(env, NULL, ".velocity <- must be as long as .position"); return NULL; }
    const size_t m = (p[0] ? len[0] : len[1]) / 3, have = g_box->count[sp];
    if (first < 0 || stride < 1 || (m && ((size_t)first >= have || (m - 1) > (have - 1 - (size_t)first) / (size_t)stride))) {
        napi_throw_range_error(env, NULL, ".position <- range lies outside the species");
        return NULL;
    }
    if (fpic_get_particles_range(h, sp, (uint64_t)first, m, (uint64_t)stride, p[0], p[1], dtype) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* commInfo(h) -> { rank, world } of the communicator the handle has joined ({0, 1} without one) */
static napi_value n_comm_info(napi_env env, napi_callback_info info)
{
    napi_value argv[1]; fpic_handle* h;
    if (!get_args(env, info, 1, argv, &h)) return NULL;
    int rank = 0, world = 1;
    if (fpic_comm_info(h, &rank, &world) != FPIC_OK) return throw_fpic(env, h);
    napi_value out, v;
    napi_create_object(env, &out);
    napi_create_int32(env, rank, &v); napi_set_named_property(env, out, "rank", v);
    napi_create_int32(env, world, &v); napi_set_named_property(env, out, "world", v);
    return out;
}

static napi_value n_get_cells_of(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fpic_handle* h; int sp;
    if (!get_args(env, info, 3, argv, &h)) return NULL;
    if (!get_species(env, argv[1], &sp)) return NULL;
    napi_typedarray_type t; void* p; size_t len;
    if (!get_typed(env, argv[2], &t, &p, &len)) return NULL;
    if (!p || t != napi_int32_array) { napi_throw_type_error(env, NULL, "expected an Int32Array"); return NULL; }
    if (!check_len(env, "cells", len, g_box->count[sp])) return NULL;
    if (fpic_get_cells_of(h, sp, (int32_t*)p) != FPIC_OK) return throw_fpic(env, h);
    return argv[2];
}

static napi_value n_add_b(napi_env env, napi_callback_info info)
{
    napi_value argv[4]; fpic_handle* h; double x, y, z;
    if (!get_args(env, info, 4, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &x) || !get_double(env, argv[2], &y) || !get_double(env, argv[3], &z)) return NULL;
    if (fpic_add_b(h, x, y, z) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* setField3(h, which, data, nx, ny, nz) */
static napi_value n_set_field3(napi_env env, napi_callback_info info)
{
    napi_value argv[6]; fpic_handle* h; double which, nx, ny, nz;
    if (!get_args(env, info, 6, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &which) || !get_double(env, argv[3], &nx) || !get_double(env, argv[4], &ny) || !get_double(env, argv[5], &nz)) return NULL;
    napi_typedarray_type t; void* data; size_t len; int dtype;
    if (!get_typed(env, argv[2], &t, &data, &len)) return NULL;
    if (!data) { napi_throw_type_error(env, NULL, "expected a typed array"); return NULL; }
    if (!float_dtype(env, t, &dtype)) return NULL;
    if (!check_len(env, "E", len, (size_t)(nx * ny * nz * 3))) return NULL;
    if (fpic_set_field3(h, (int)which, data, (int)nx, (int)ny, (int)nz, dtype) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* readField3(h, which, out): Float32Array / Float64Array, or BigInt64Array for the fixed-point charge grid */
static napi_value n_read_field3(napi_env env, napi_callback_info info)
{
    napi_value argv[3]; fpic_handle* h; double which;
    if (!get_args(env, info, 3, argv, &h)) return NULL;
    if (!get_double(env, argv[1], &which)) return NULL;
    napi_typedarray_type t; void* data; size_t len; int dtype = FPIC_F32;
    if (!get_typed(env, argv[2], &t, &data, &len)) return NULL;
    if (!data) { napi_throw_type_error(env, NULL, "expected a typed array"); return NULL; }
    if ((int)which == FPIC_F3_RHO_FIXED || (int)which == FPIC_F3_J_FIXED) {
        if (t != napi_bigint64_array) { napi_throw_type_error(env, NULL, "expected a BigInt64Array"); return NULL; }
    } else if (!float_dtype(env, t, &dtype)) return NULL;
    if (!check_len(env, "out", len, g_box->nodes * (((int)which == FPIC_F3_RHO || (int)which == FPIC_F3_PHI || (int)which == FPIC_F3_RHO_FIXED) ? 1 : ((int)which == FPIC_F3_J_FIXED ? 3 : 4)))) return NULL;
    if (fpic_read_field3(h, (int)which, data, dtype) != FPIC_OK) return throw_fpic(env, h);
    return argv[2];
}

/* ---- multi-GPU: the library's RCCL communicator ---- */
static napi_value n_comm_unique_id(napi_env env, napi_callback_info info)
{
    (void)info;
    void* data = NULL; napi_value ab, out;
    NAPI_OK(env, napi_create_arraybuffer(env, FPIC_UNIQUE_ID_BYTES, &data, &ab));
    if (fpic_comm_unique_id(data) != FPIC_OK) return throw_fpic(env, NULL);
    NAPI_OK(env, napi_create_typedarray(env, napi_uint8_array, FPIC_UNIQUE_ID_BYTES, ab, 0, &out));
    return out;
}

/* commInit(h, id Uint8Array(128), rank, world, overlap) */
static napi_value n_comm_init(napi_env env, napi_callback_info info)
{
    napi_value argv[5]; fpic_handle* h; double rank, world, overlap;
    if (!get_args(env, info, 5, argv, &h)) return NULL;
    napi_typedarray_type t; void* id; size_t len;
    if (!get_typed(env, argv[1], &t, &id, &len)) return NULL;
    if (!id || t != napi_uint8_array) { napi_throw_type_error(env, NULL, "expected a Uint8Array"); return NULL; }
    if (!check_len(env, "id", len, FPIC_UNIQUE_ID_BYTES)) return NULL;
    if (!get_double(env, argv[2], &rank) || !get_double(env, argv[3], &world) || !get_double(env, argv[4], &overlap)) return NULL;
    if (fpic_comm_init(h, id, (int)rank, (int)world) != FPIC_OK) return throw_fpic(env, h);
    if (fpic_comm_set_overlap(h, (int)overlap) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}
SIMPLE_CALL(n_comm_destroy, fpic_comm_destroy)

/* ---- multi-GPU: z-slab decomposition of the box (this process = one rank; exchange over the communicator) ---- */
/* domainInit(h, rank, world, ghostPlanes, migrateEvery, distributedSolve) */
static napi_value n_domain_init(napi_env env, napi_callback_info info)
{
    napi_value argv[6]; fpic_handle* h; double v[5];
    if (!get_args(env, info, 6, argv, &h)) return NULL;
    for (int k = 0; k < 5; ++k)
        if (!get_double(env, argv[1 + k], &v[k])) return NULL;
    if (fpic_domain_init(h, (int)v[0], (int)v[1], (int)v[2], (int)v[3], (int)v[4]) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* domainSetParticles(h, species, position, velocity, firstId): the rank's initial particles, global indices firstId.. */
static napi_value n_domain_set_particles(napi_env env, napi_callback_info info)
{
    napi_value argv[5]; fpic_handle* h; int sp; double first;
    if (!get_args(env, info, 5, argv, &h)) return NULL;
    if (!get_species(env, argv[1], &sp) || !get_double(env, argv[4], &first)) return NULL;
    napi_typedarray_type t[2]; void* p[2]; size_t len[2]; int d[2];
    for (int k = 0; k < 2; ++k) {
        if (!get_typed(env, argv[2 + k], &t[k], &p[k], &len[k])) return NULL;
        if (!p[k]) { napi_throw_type_error(env, NULL, "expected position and velocity typed arrays"); return NULL; }
        if (!float_dtype(env, t[k], &d[k])) return NULL;
    }
    if (d[0] != d[1]) { napi_throw_type_error(env, NULL, "position and velocity must have the same element type"); return NULL; }
    if (len[0] % 3) { napi_throw_range_error(env, NULL, ".position <- length must be a multiple of 3"); return NULL; }
    if (!check_len(env, "velocity", len[1], len[0])) return NULL;
    if (len[0] / 3 > g_box->count[sp]) { napi_throw_range_error(env, NULL, ".position <- more particles than the species' capacity"); return NULL; }
    if (first < 0 || first > 4294967295.0) { napi_throw_range_error(env, NULL, ".firstId <- must fit 32 bits"); return NULL; }
    if (fpic_domain_set_particles(h, sp, len[0] / 3, p[0], p[1], (uint32_t)first, d[0]) != FPIC_OK) return throw_fpic(env, h);
    return undefined(env);
}

/* domainGetParticles(h, species, position, velocity, ids) -> count; all three sized for the species' capacity */
static napi_value n_domain_get_particles(napi_env env, napi_callback_info info)
{
    napi_value argv[5]; fpic_handle* h; int sp;
    if (!get_args(env, info, 5, argv, &h)) return NULL;
    if (!get_species(env, argv[1], &sp)) return NULL;
    napi_typedarray_type t[3]; void* p[3]; size_t len[3]; int d[2];
    for (int k = 0; k < 3; ++k) {
        if (!get_typed(env, argv[2 + k], &t[k], &p[k], &len[k])) return NULL;
        if (!p[k]) { napi_throw_type_error(env, NULL, "expected position, velocity and ids typed arrays"); return NULL; }
    }
    if (!float_dtype(env, t[0], &d[0]) || !float_dtype(env, t[1], &d[1])) return NULL;
    if (d[0] != d[1]) { napi_throw_type_error(env, NULL, "position and velocity must have the same element type"); return NULL; }
    if (t[2] != napi_uint32_array) { napi_throw_type_error(env, NULL, "expected a Uint32Array of ids"); return NULL; }
    if (!check_len(env, "position", len[0], 3 * g_box->count[sp]) || !check_len(env, "velocity", len[1], 3 * g_box->count[sp]) ||
        !check_len(env, "ids", len[2], g_box->count[sp]))
        return NULL;
    uint64_t n = 0;
    if (fpic_domain_get_particles(h, sp, p[0], p[1], (uint32_t*)p[2], g_box->count[sp], &n, d[0]) != FPIC_OK) return throw_fpic(env, h);
    napi_value out;
    NAPI_OK(env, napi_create_double(env, (double)n, &out));
    return out;
}

static napi_value n_domain_stats(napi_env env, napi_callback_info info)
{
    napi_value argv[1]; fpic_handle* h;
    if (!get_args(env, info, 1, argv, &h)) return NULL;
    uint64_t migrated = 0, lost = 0;
    if (fpic_domain_stats(h, &migrated, &lost) != FPIC_OK) return throw_fpic(env, h);
    napi_value out, a, b;
    NAPI_OK(env, napi_create_object(env, &out));
    NAPI_OK(env, napi_create_double(env, (double)migrated, &a));
    NAPI_OK(env, napi_create_double(env, (double)lost, &b));
    NAPI_OK(env, napi_set_named_property(env, out, "migrated", a));
    NAPI_OK(env, napi_set_named_property(env, out, "lost", b));
    return out;
}

static napi_value n_build_arch(napi_env env, napi_callback_info info)
{
    (void)info;
    napi_value s;
    NAPI_OK(env, napi_create_string_utf8(env, fpic_build_arch(), NAPI_AUTO_LENGTH, &s));
    return s;
}

int fusionsor_register(napi_env env, napi_value exports); /* fusionsor_napi.c */

static napi_value init(napi_env env, napi_value exports)
{
    struct { const char* name; napi_callback fn; } table[] = {
        { "create", n_create }, { "destroy", n_destroy }, { "setParticles", n_set_particles },
        { "setGrid", n_set_grid }, { "setRandomState", n_set_random_state }, { "addCurrentLoop", n_add_current_loop },
        { "addCurrentZ", n_add_current_z }, { "addBZ", n_add_bz }, { "addBTheta", n_add_btheta },
        { "precalc", n_precalc }, { "step", n_step }, { "substeps", n_substeps }, { "stepAsync", n_step_async }, { "density", n_density }, { "deposit", n_deposit },
        { "densityFinish", n_density_finish }, { "readGrid", n_read_grid }, { "getParticles", n_get_particles },
        { "getCells", n_get_cells }, { "sort", n_sort }, { "sync", n_sync }, { "profile", n_profile },
        { "getStats", n_get_stats }, { "saveCheckpoint", n_save_checkpoint }, { "loadCheckpoint", n_load_checkpoint }, { "resetStats", n_reset_stats }, { "buildArch", n_build_arch },
        { "addSpecies", n_add_species }, { "setParticlesRange", n_set_particles_range }, { "getParticlesOf", n_get_particles_of },
        { "getCellsOf", n_get_cells_of }, { "getParticlesRange", n_get_particles_range }, { "commInfo", n_comm_info }, { "addB", n_add_b }, { "setField3", n_set_field3 }, { "readField3", n_read_field3 },
        { "commUniqueId", n_comm_unique_id }, { "commInit", n_comm_init }, { "commDestroy", n_comm_destroy },
        { "domainInit", n_domain_init }, { "domainSetParticles", n_domain_set_particles }, { "domainGetParticles", n_domain_get_particles },
        { "domainStats", n_domain_stats },
    };
    for (size_t i = 0; i < sizeof table / sizeof table[0]; ++i) {
        napi_value fn;
        if (napi_create_function(env, table[i].name, NAPI_AUTO_LENGTH, table[i].fn, NULL, &fn) != napi_ok ||
            napi_set_named_property(env, exports, table[i].name, fn) != napi_ok) {
            napi_throw_error(env, NULL, "addon initialisation failed");
            return NULL;
        }
    }
    if (!fusionsor_register(env, exports)) {
        napi_throw_error(env, NULL, "addon initialisation failed");
        return NULL;
    }
    return exports;
}

#ifndef NODE_GYP_MODULE_NAME
#define NODE_GYP_MODULE_NAME fusionpic_napi
#endif
NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
