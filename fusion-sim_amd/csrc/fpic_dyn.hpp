// fpic_dyn.hpp — rocFFT and RCCL bound at run time (dlopen), so that libfusionpic.so itself links
// only the HIP runtime: a host that never asks for the FFT field solve or for a multi-GPU
// communicator (the plain reference-parity pusher under Node) needs neither library, and inside a
// process that has already loaded them (PyTorch-ROCm ships its own copies with the same sonames)
// the loaded instance is the one that is used.  Not part of the ABI.
#pragma once

#include <rccl/rccl.h>
#include <rocfft/rocfft.h>

#include <string>

namespace fdyn {

struct RocFFT {
    bool ok = false;
    std::string why;
    decltype(&rocfft_setup) setup = nullptr;
    decltype(&rocfft_plan_create) plan_create = nullptr;
    decltype(&rocfft_plan_destroy) plan_destroy = nullptr;
    decltype(&rocfft_plan_description_create) plan_description_create = nullptr;
    decltype(&rocfft_plan_description_destroy) plan_description_destroy = nullptr;
    decltype(&rocfft_plan_description_set_data_layout) plan_description_set_data_layout = nullptr;
    decltype(&rocfft_plan_get_work_buffer_size) plan_get_work_buffer_size = nullptr;
    decltype(&rocfft_execution_info_create) execution_info_create = nullptr;
    decltype(&rocfft_execution_info_destroy) execution_info_destroy = nullptr;
    decltype(&rocfft_execution_info_set_work_buffer) execution_info_set_work_buffer = nullptr;
    decltype(&rocfft_execution_info_set_stream) execution_info_set_stream = nullptr;
    decltype(&rocfft_execute) execute = nullptr;
};

struct Rccl {
    bool ok = false;
    std::string why;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const RocFFT& rocfft(); // loaded (and rocfft_setup called) on first use
const Rccl& rccl();

} // namespace fdyn
