// fpic_dyn.cpp — see fpic_dyn.hpp.
#include "fpic_dyn.hpp"

#include <dlfcn.h>

#include <mutex>

namespace fdyn {
namespace {

void* open_first(const char* const* names, std::string& why)
{
    for (const char* const* n = names; *n; ++n) {
        // RTLD_NOLOAD first: an instance the process already holds (same soname) wins
        if (void* h = dlopen(*n, RTLD_NOW | RTLD_NOLOAD)) return h;
    }
    for (const char* const* n = names; *n; ++n) {
        if (void* h = dlopen(*n, RTLD_NOW | RTLD_LOCAL)) return h;
        const char* e = dlerror();
        why += std::string(*n) + ": " + (e ? e : "dlopen failed") + "; ";
    }
    return nullptr;
}

template <typename F>
bool bind(void* lib, const char* name, F& fn, std::string& why)
{
    fn = reinterpret_cast<F>(dlsym(lib, name));
    if (!fn) why += std::string("missing symbol ") + name + "; ";
    return fn != nullptr;
}

} // namespace

const RocFFT& rocfft()
{
    static RocFFT r;
    static std::once_flag once;
    std::call_once(once, [] {
        static const char* const names[] = { "librocfft.so.0", "librocfft.so", "/opt/rocm/lib/librocfft.so.0", nullptr };
        void* lib = open_first(names, r.why);
        if (!lib) return;
        bool ok = bind(lib, "rocfft_setup", r.setup, r.why);
        ok &= bind(lib, "rocfft_plan_create", r.plan_create, r.why);
        ok &= bind(lib, "rocfft_plan_destroy", r.plan_destroy, r.why);
        ok &= bind(lib, "rocfft_plan_description_create", r.plan_description_create, r.why);
        ok &= bind(lib, "rocfft_plan_description_destroy", r.plan_description_destroy, r.why);
        ok &= bind(lib, "rocfft_plan_description_set_data_layout", r.plan_description_set_data_layout, r.why);
        ok &= bind(lib, "rocfft_plan_get_work_buffer_size", r.plan_get_work_buffer_size, r.why);
        ok &= bind(lib, "rocfft_execution_info_create", r.execution_info_create, r.why);
        ok &= bind(lib, "rocfft_execution_info_destroy", r.execution_info_destroy, r.why);
        ok &= bind(lib, "rocfft_execution_info_set_work_buffer", r.execution_info_set_work_buffer, r.why);
        ok &= bind(lib, "rocfft_execution_info_set_stream", r.execution_info_set_stream, r.why);
        ok &= bind(lib, "rocfft_execute", r.execute, r.why);
        if (ok && r.setup() != rocfft_status_success) { r.why += "rocfft_setup failed; "; ok = false; }
        r.ok = ok;
    });
    return r;
}

const Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        static const char* const names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr };
        void* lib = nullptr;
        // FPIC_RCCL_LIBRARY names the RCCL build to bind (the tests bind an in-process stand-in, tests/fake_rccl)
        if (const char* over = std::getenv("FPIC_RCCL_LIBRARY"); over && *over) {
            lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!lib) { const char* e = dlerror(); r.why = std::string(over) + ": " + (e ? e : "dlopen failed"); }
        } else {
            lib = open_first(names, r.why);
        }
        if (!lib) return;
        bool ok = bind(lib, "ncclGetUniqueId", r.GetUniqueId, r.why);
        ok &= bind(lib, "ncclCommInitRank", r.CommInitRank, r.why);
        ok &= bind(lib, "ncclCommDestroy", r.CommDestroy, r.why);
        ok &= bind(lib, "ncclAllReduce", r.AllReduce, r.why);
        ok &= bind(lib, "ncclAllGather", r.AllGather, r.why);
        ok &= bind(lib, "ncclSend", r.Send, r.why);
        ok &= bind(lib, "ncclRecv", r.Recv, r.why);
        ok &= bind(lib, "ncclGroupStart", r.GroupStart, r.why);
        ok &= bind(lib, "ncclGroupEnd", r.GroupEnd, r.why);
        ok &= bind(lib, "ncclGetErrorString", r.GetErrorString, r.why);
        r.ok = ok;
    });
    return r;
}

} // namespace fdyn
