// fsor_api.hip — extern "C" entry points of include/fusionsor.h: the dense iterative solver
// of the reference (matrix_webgl.js makeSORIterative) on gfx950.  Kernels: fsor_kernels.hpp.
#include "../../include/fusionsor.h"
#include "fpic_internal.hpp"
#include "fsor_kernels.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

using namespace fsor;

namespace {
thread_local std::string g_sor_create_error;
}

struct fsor_handle {
    fsor_spec spec{};
    Shape shape{};
    int device = 0;
    double omega = 1.0;
    bool relaxed = false; // omega !== 1.0: R and C are scaled and the (1-omega) x term exists
    float omega_lit = 1.0f, keep_lit = 0.0f;
    float* A = nullptr;   // L*L, row-major
    float4* R = nullptr;  // L*T texels, storage order of fsor_kernels.hpp
    float* b = nullptr;
    float* C = nullptr;
    float* x[2] = {nullptr, nullptr};  // natural order; x[cur] = x_result, x[cur^1] = x_guess
    float* xs[2] = {nullptr, nullptr}; // the same vectors in (slot, lane) order
    float* stats = nullptr;
    int cur = 0;
    bool have_matrix = false, have_b = false, prepared = false;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool profiling = false;
    fsor_stats st{};
    std::vector<float> host_a, host_b, host_c; // read-back staging of solve()
    std::string err;
};

namespace {

int fail(fsor_handle* h, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_sor_create_error = buf;
    return code;
}

#define SOR_TRY(h, expr)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(h, e_ == hipErrorOutOfMemory ? FPIC_ERR_OOM : FPIC_ERR_HIP, "%s failed: %s", #expr,       \
                        hipGetErrorString(e_));                                                                   \
    } while (0)

#define SOR_HANDLE(h)                                                                                             \
    do {                                                                                                          \
        if (!(h)) return fail(nullptr, FPIC_ERR_INVALID_ARG, "null handle");                                      \
        hipError_t e_ = hipSetDevice((h)->device);                                                                \
        if (e_ != hipSuccess) return fail(h, FPIC_ERR_HIP, "hipSetDevice failed: %s", hipGetErrorString(e_));     \
    } while (0)

void release(fsor_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : {static_cast<void*>(h->A), static_cast<void*>(h->R), static_cast<void*>(h->b), static_cast<void*>(h->C),
                    static_cast<void*>(h->x[0]), static_cast<void*>(h->x[1]), static_cast<void*>(h->xs[0]),
                    static_cast<void*>(h->xs[1]), static_cast<void*>(h->stats)})
        if (p) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

// host array (float or double) -> device floats, one rounding (a Float32Array store)
int upload(fsor_handle* h, float* dst, const void* src, size_t n, int dtype)
{
    if (!src) return fail(h, FPIC_ERR_INVALID_ARG, "null host pointer");
    if (dtype == FPIC_F32) {
        SOR_TRY(h, hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
        SOR_TRY(h, hipStreamSynchronize(h->stream));
        return FPIC_OK;
    }
    if (dtype != FPIC_F64) return fail(h, FPIC_ERR_INVALID_ARG, "dtype must be FPIC_F32 or FPIC_F64");
    // convert in bounded chunks so a 2^32-element matrix does not need a second full host copy
    const size_t chunk = size_t(1) << 24;
    std::vector<float> tmp(n < chunk ? n : chunk);
    const double* s = static_cast<const double*>(src);
    for (size_t at = 0; at < n; at += chunk) {
        const size_t m = n - at < chunk ? n - at : chunk;
        for (size_t i = 0; i < m; ++i) tmp[i] = static_cast<float>(s[at + i]);
        SOR_TRY(h, hipMemcpyAsync(dst + at, tmp.data(), m * sizeof(float), hipMemcpyHostToDevice, h->stream));
        SOR_TRY(h, hipStreamSynchronize(h->stream));
    }
    return FPIC_OK;
}

int launch_permute(fsor_handle* h, int which)
{
    const Shape& s = h->shape;
    permute_vector_kernel<<<(s.T + 255) / 256, 256, 0, h->stream>>>(reinterpret_cast<const float4*>(h->x[which]),
                                                                   reinterpret_cast<float4*>(h->xs[which]), s);
    SOR_TRY(h, hipGetLastError());
    return FPIC_OK;
}

int launch_product(fsor_handle* h)
{
    const Shape& s = h->shape;
    const int in = h->cur, out = h->cur ^ 1; // x_guess <- x_result is a swap of roles, not a copy
    const dim3 grid((s.L + kRowsPerBlock - 1) / kRowsPerBlock), block(kWave * kRowsPerBlock);
    const float4* R = h->R;
    const float4* xs_in = reinterpret_cast<const float4*>(h->xs[in]);
#define FSOR_LAUNCH(LOCAL)                                                                                        \
    product_kernel<LOCAL><<<grid, block, 0, h->stream>>>(R, xs_in, h->x[in], h->C, h->x[out], h->xs[out], s,      \
                                                         h->keep_lit, h->relaxed ? 1 : 0, h->spec.natural_rows)
    switch (s.levels_local) {
    case 0: FSOR_LAUNCH(0); break;
    case 1: FSOR_LAUNCH(1); break;
    case 2: FSOR_LAUNCH(2); break;
    case 3: FSOR_LAUNCH(3); break;
    case 4: FSOR_LAUNCH(4); break;
    default: return fail(h, FPIC_ERR_INVALID_ARG, ".n_power <- unsupported size");
    }
#undef FSOR_LAUNCH
    SOR_TRY(h, hipGetLastError());
    h->cur = out;
    h->st.iterations++;
    return FPIC_OK;
}

} // namespace

extern "C" {

const char* fsor_last_error(const fsor_handle* h) { return h ? h->err.c_str() : g_sor_create_error.c_str(); }
int fsor_abi_version(void) { return FSOR_ABI_VERSION; }

int fsor_create(const fsor_spec* spec, fsor_handle** out)
{
    if (!spec || !out) return fail(nullptr, FPIC_ERR_INVALID_ARG, "null spec or output pointer");
    *out = nullptr;
    // n_power = 0 links programResult against sum_buffers[-1] in the reference, which throws
    // (matrix_webgl.js:420-424, utilities.js:333-335)
    if (spec->n_power == 0) return fail(nullptr, FPIC_ERR_INVALID_ARG, "Cannot add uniform value: u_Vsum");
    if (spec->n_power < 1 || spec->n_power > 7)
        return fail(nullptr, FPIC_ERR_INVALID_ARG, ".n_power <- %d outside 1..7 (the matrix holds 16^(n_power+1) floats)", spec->n_power);
    if (!(spec->relaxation == spec->relaxation)) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".relaxation <- NaN");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, FPIC_ERR_NO_DEVICE, "no HIP device visible: libfusionpic.so has no CPU fallback");
    if (spec->device < 0 || spec->device >= ndev) return fail(nullptr, FPIC_ERR_INVALID_ARG, ".device <- %d of %d", spec->device, ndev);

    fsor_handle* h = new fsor_handle;
    h->spec = *spec;
    h->device = spec->device;
    h->shape = make_shape(spec->n_power);
    h->omega = spec->relaxation != 0.0 ? spec->relaxation : 1.0; // spec.relaxation || 1.0
    h->relaxed = h->omega != 1.0;
    h->omega_lit = static_cast<float>(fpic::shader_literal(h->omega));        // N(omega), matrix_webgl.js:255, :294
    h->keep_lit = static_cast<float>(fpic::shader_literal(1.0 - h->omega));   // N(1.0 - omega), :412
    const size_t L = h->shape.L;
    h->st.matrix_bytes = L * L * sizeof(float);

    auto bail = [&](int code) { g_sor_create_error = h->err; release(h); return code; };
    hipError_t e;
    if (hipSetDevice(h->device) != hipSuccess) return bail(fail(h, FPIC_ERR_HIP, "hipSetDevice failed"));
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(fail(h, FPIC_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e)));
    h->stream = h->own_stream;
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess)
        return bail(fail(h, FPIC_ERR_HIP, "hipEventCreate failed"));
    struct { void** p; size_t bytes; } allocs[] = {
        {reinterpret_cast<void**>(&h->A), L * L * sizeof(float)}, {reinterpret_cast<void**>(&h->R), L * L * sizeof(float)},
        {reinterpret_cast<void**>(&h->b), L * sizeof(float)},     {reinterpret_cast<void**>(&h->C), L * sizeof(float)},
        {reinterpret_cast<void**>(&h->x[0]), L * sizeof(float)},  {reinterpret_cast<void**>(&h->x[1]), L * sizeof(float)},
        {reinterpret_cast<void**>(&h->xs[0]), L * sizeof(float)}, {reinterpret_cast<void**>(&h->xs[1]), L * sizeof(float)},
        {reinterpret_cast<void**>(&h->stats), L * sizeof(float)},
    };
    for (auto& a : allocs) {
        if ((e = hipMalloc(a.p, a.bytes)) != hipSuccess)
            return bail(fail(h, e == hipErrorOutOfMemory ? FPIC_ERR_OOM : FPIC_ERR_HIP, "hipMalloc of %zu bytes failed: %s", a.bytes,
                             hipGetErrorString(e)));
        // frame buffers and Float32Arrays start at zero (utilities.js:533-539)
        if ((e = hipMemsetAsync(*a.p, 0, a.bytes, h->stream)) != hipSuccess)
            return bail(fail(h, FPIC_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e)));
    }
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail(fail(h, FPIC_ERR_HIP, "%s", hipGetErrorString(e)));
    h->host_a.resize(L);
    h->host_b.resize(L);
    h->host_c.resize(L);
    *out = h;
    return FPIC_OK;
}

void fsor_destroy(fsor_handle* h) { release(h); }

int fsor_dims(const fsor_handle* h, uint64_t* vec_length, uint32_t* vec_height)
{
    if (!h) return fail(nullptr, FPIC_ERR_INVALID_ARG, "null handle");
    if (vec_length) *vec_length = h->shape.L;
    if (vec_height) *vec_height = static_cast<uint32_t>(h->shape.vh);
    return FPIC_OK;
}

int fsor_set_matrix(fsor_handle* h, const void* a, int dtype)
{
    SOR_HANDLE(h);
    const size_t L = h->shape.L;
    if (int rc = upload(h, h->A, a, L * L, dtype)) return rc;
    h->have_matrix = true;
    h->prepared = false;
    return FPIC_OK;
}

int fsor_set_b(fsor_handle* h, const void* b, int dtype)
{
    SOR_HANDLE(h);
    if (int rc = upload(h, h->b, b, h->shape.L, dtype)) return rc;
    h->have_b = true;
    h->prepared = false;
    return FPIC_OK;
}

int fsor_init_vector(fsor_handle* h, const void* x, int dtype)
{
    SOR_HANDLE(h);
    if (int rc = upload(h, h->x[h->cur], x, h->shape.L, dtype)) return rc;
    return launch_permute(h, h->cur);
}

int fsor_prepare(fsor_handle* h)
{
    SOR_HANDLE(h);
    // the reference would sample a null texture here and throw inside gl; say what is missing
    if (!h->have_matrix) return fail(h, FPIC_ERR_STATE, "solve before set_matrix");
    if (!h->have_b) return fail(h, FPIC_ERR_STATE, "solve before set_b");
    const Shape& s = h->shape;
    const size_t texels = static_cast<size_t>(s.T) * s.L;
    build_iteration_matrix_kernel<<<static_cast<unsigned>((texels + 255) / 256), 256, 0, h->stream>>>(h->A, h->R, s, h->omega_lit,
                                                                                                   h->relaxed ? 1 : 0);
    SOR_TRY(h, hipGetLastError());
    build_constant_kernel<<<(s.L + 255) / 256, 256, 0, h->stream>>>(h->A, h->b, h->C, s.L, h->omega_lit, h->relaxed ? 1 : 0);
    SOR_TRY(h, hipGetLastError());
    h->prepared = true;
    return FPIC_OK;
}

int fsor_iterate(fsor_handle* h, int32_t n)
{
    SOR_HANDLE(h);
    if (!h->prepared) return fail(h, FPIC_ERR_STATE, "iterate before prepare");
    if (n < 0) return fail(h, FPIC_ERR_INVALID_ARG, "negative iteration count");
    if (h->profiling) SOR_TRY(h, hipEventRecord(h->ev0, h->stream));
    for (int32_t k = 0; k < n; ++k)
        if (int rc = launch_product(h)) return rc;
    if (h->profiling) {
        SOR_TRY(h, hipEventRecord(h->ev1, h->stream));
        SOR_TRY(h, hipEventSynchronize(h->ev1));
        float ms = 0.f;
        SOR_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->st.seconds_iterate += 1e-3 * ms;
    }
    return FPIC_OK;
}

int fsor_solve(fsor_handle* h, double tolerance, int32_t substep, int32_t has_max, int32_t max_iterations, fsor_result* out,
               float* result)
{
    SOR_HANDLE(h);
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, "null result pointer");
    if (int rc = fsor_prepare(h)) return rc;
    const Shape& s = h->shape;
    const size_t L = s.L, n_vec = static_cast<size_t>(s.T);
    const size_t bytes = L * sizeof(float);
    // matrix_webgl.js:592: the debug read-back leaves C in the closure array that is returned
    SOR_TRY(h, hipMemcpyAsync(h->host_b.data(), h->C, bytes, hipMemcpyDeviceToHost, h->stream));
    SOR_TRY(h, hipStreamSynchronize(h->stream));

    double correlation = 0.0, x1 = 0, x2 = 0, x1x2 = 0, x1x1 = 0, x2x2 = 0; // never reset inside the loop (:613-618)
    double diff = tolerance + 1;
    int iteration = 0;
    if (substep <= 0) substep = 1;
    while (has_max && iteration < max_iterations && diff > tolerance) {
        if (int rc = fsor_iterate(h, substep)) return rc;
        const float* guess = h->x[h->cur ^ 1];
        const float* res = h->x[h->cur];
        stats_kernel<<<(s.T + 255) / 256, 256, 0, h->stream>>>(reinterpret_cast<const float4*>(guess), reinterpret_cast<const float4*>(res),
                                                               reinterpret_cast<float4*>(h->stats), static_cast<uint32_t>(s.T));
        SOR_TRY(h, hipGetLastError());
        SOR_TRY(h, hipMemcpyAsync(h->host_c.data(), h->stats, bytes, hipMemcpyDeviceToHost, h->stream));
        SOR_TRY(h, hipMemcpyAsync(h->host_a.data(), guess, bytes, hipMemcpyDeviceToHost, h->stream));
        SOR_TRY(h, hipMemcpyAsync(h->host_b.data(), res, bytes, hipMemcpyDeviceToHost, h->stream));
        SOR_TRY(h, hipStreamSynchronize(h->stream));
        const float *a = h->host_a.data(), *b = h->host_b.data(), *st = h->host_c.data();
        double max_diff = 0.0;
        for (size_t i = 0; i < n_vec; ++i) { // matrix_webgl.js:660-669, JavaScript doubles
            x1 += ((static_cast<double>(a[4 * i]) + a[4 * i + 1]) + a[4 * i + 2]) + a[4 * i + 3];
            x2 += ((static_cast<double>(b[4 * i]) + b[4 * i + 1]) + b[4 * i + 2]) + b[4 * i + 3];
            x1x2 += st[4 * i];
            x1x1 += st[4 * i + 1];
            x2x2 += st[4 * i + 2];
            const double d = st[4 * i + 3];
            max_diff = (max_diff != max_diff || d != d) ? std::numeric_limits<double>::quiet_NaN() : (max_diff < d ? d : max_diff);
        }
        const double n = static_cast<double>(L);
        correlation = (n * x1x2 - x1 * x2) / std::sqrt((n * x1x1 - x1 * x1) * (n * x2x2 - x2 * x2));
        diff = 2 * n * max_diff / (std::fabs(x1) + std::fabs(x2));
        iteration++;
    }
    out->correlation = correlation;
    out->diff = diff;
    out->iterations = iteration;
    out->reserved = 0;
    if (result) std::memcpy(result, h->host_b.data(), bytes);
    return FPIC_OK;
}

int fsor_read_vector(fsor_handle* h, int which, float* out)
{
    SOR_HANDLE(h);
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, "null output pointer");
    const float* src = nullptr;
    switch (which) {
    case FSOR_X_RESULT: src = h->x[h->cur]; break;
    case FSOR_X_GUESS: src = h->x[h->cur ^ 1]; break;
    case FSOR_X_STATS: src = h->stats; break;
    case FSOR_C: src = h->C; break;
    case FSOR_B: src = h->b; break;
    default: return fail(h, FPIC_ERR_INVALID_ARG, "unknown vector %d", which);
    }
    SOR_TRY(h, hipMemcpyAsync(out, src, h->shape.L * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    SOR_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

int fsor_read_iteration_matrix(fsor_handle* h, float* out)
{
    SOR_HANDLE(h);
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, "null output pointer");
    if (!h->prepared) return fail(h, FPIC_ERR_STATE, "iteration matrix read before prepare");
    const Shape& s = h->shape;
    const size_t bytes = static_cast<size_t>(s.L) * s.L * sizeof(float);
    float4* tex = nullptr;
    SOR_TRY(h, hipMalloc(reinterpret_cast<void**>(&tex), bytes));
    const size_t texels = bytes / sizeof(float4);
    export_iteration_matrix_kernel<<<static_cast<unsigned>((texels + 255) / 256), 256, 0, h->stream>>>(h->R, tex, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, tex, bytes, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tex);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "iteration matrix read-back failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

int fsor_device_buffer(fsor_handle* h, int which, void** ptr, size_t* bytes)
{
    SOR_HANDLE(h);
    if (!ptr || !bytes) return fail(h, FPIC_ERR_INVALID_ARG, "null output pointer");
    switch (which) {
    case FSOR_X_RESULT: *ptr = h->x[h->cur]; break;
    case FSOR_X_GUESS: *ptr = h->x[h->cur ^ 1]; break;
    case FSOR_X_STATS: *ptr = h->stats; break;
    case FSOR_C: *ptr = h->C; break;
    case FSOR_B: *ptr = h->b; break;
    default: return fail(h, FPIC_ERR_INVALID_ARG, "unknown vector %d", which);
    }
    *bytes = h->shape.L * sizeof(float);
    return FPIC_OK;
}

int fsor_set_stream(fsor_handle* h, void* hip_stream)
{
    SOR_HANDLE(h);
    SOR_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return FPIC_OK;
}

int fsor_sync(fsor_handle* h)
{
    SOR_HANDLE(h);
    SOR_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

int fsor_profile(fsor_handle* h, int enable)
{
    SOR_HANDLE(h);
    h->profiling = enable != 0;
    return FPIC_OK;
}

int fsor_get_stats(fsor_handle* h, fsor_stats* out)
{
    SOR_HANDLE(h);
    if (!out) return fail(h, FPIC_ERR_INVALID_ARG, "null output pointer");
    *out = h->st;
    return FPIC_OK;
}

int fsor_reset_stats(fsor_handle* h)
{
    SOR_HANDLE(h);
    const uint64_t bytes = h->st.matrix_bytes;
    h->st = fsor_stats{};
    h->st.matrix_bytes = bytes;
    return FPIC_OK;
}

} // extern "C"
