// fpic_comm.hpp — the multi-GPU side of libfusionpic.so: one process per GPU, one handle per process,
// RCCL (bound at run time, fpic_dyn.hpp) over xGMI.  The reference has no communication at all (one
// WebGL context); the shapes follow SURVEY.md 8(e).  Not part of the ABI.
#pragma once

#include "fpic_handle.hpp"
#include "fpic_dyn.hpp"

namespace fcomm {

struct Comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1;
    bool overlap = true;
    // reference-parity mode: the frame's single exchange runs beside the next push
    hipStream_t side = nullptr;
    void* buf = nullptr;
    size_t buf_bytes = 0;
    hipEvent_t copied = nullptr, reduced = nullptr;
    bool reduce_pending = false;
};

int check(fpic_handle* h, ncclResult_t r, const char* what);
void release(fpic_handle* h);

} // namespace fcomm
