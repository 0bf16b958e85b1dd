// fes_host_io.inc.hpp: uploads and downloads of particles and grids, creation of the state — part of fes_api.hip's translation unit (included there, inside namespace fes; not a header of its own:
// the pieces share the anonymous namespace's templates).  Split out in round 5 without changing a symbol.
template <typename T, typename In>
int upload_pos(fpic_handle* h, Species& s, const In* host, size_t first, size_t count)
{
    State* st = h->es;
    const size_t chunk = 8u << 20;
    In* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * 3 * sizeof(In)));
    T* a = static_cast<T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyDefault, h->stream); // host or device memory
        if (e == hipSuccess) {
            if (s.ids_identity)
                set_pos3_kernel<T, In><<<blocks_for(m), 256, 0, h->stream>>>(stage, first + b, m, 1 / st->lx, 1 / st->ly, 1 / st->lz, a, a + s.n_pad, a + 2 * s.n_pad,
                                                                           nullptr, first + b + m, first + b);
            else
                set_pos3_kernel<T, In><<<blocks_for(s.n), 256, 0, h->stream>>>(stage, first + b, m, 1 / st->lx, 1 / st->ly, 1 / st->lz, a, a + s.n_pad,
                                                                             a + 2 * s.n_pad, s.id[s.cur], s.n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle upload failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename In>
int upload_vel(fpic_handle* h, Species& s, const In* host, size_t first, size_t count)
{
    const size_t chunk = 8u << 20;
    In* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * 3 * sizeof(In)));
    T* a = static_cast<T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        hipError_t e = hipMemcpyAsync(stage, host + 3 * b, m * 3 * sizeof(In), hipMemcpyDefault, h->stream); // host or device memory
        if (e == hipSuccess) {
            // velocities stay in units of c, unscaled
            if (s.ids_identity)
                set_vec3_kernel<T, In><<<blocks_for(m), 256, 0, h->stream>>>(stage, first + b, m, 1.0, 1.0, a + 3 * s.n_pad, a + 4 * s.n_pad, a + 5 * s.n_pad,
                                                                           nullptr, nullptr, first + b + m, first + b);
            else
                set_vec3_kernel<T, In><<<blocks_for(s.n), 256, 0, h->stream>>>(stage, first + b, m, 1.0, 1.0, a + 3 * s.n_pad, a + 4 * s.n_pad,
                                                                             a + 5 * s.n_pad, nullptr, s.id[s.cur], s.n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle upload failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename Out>
int download_vec3(fpic_handle* h, const Species& s, Out* host, int first, size_t from = 0, size_t count = ~size_t(0), size_t stride = 1)
{
    // `count` of the caller's particles from, from + stride, ... (everything by default), in pieces of `chunk` output slots
    if (count == ~size_t(0)) count = s.n;
    if (!count) return FPIC_OK;
    const size_t chunk = 8u << 20;
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * 3 * sizeof(Out)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        get_vec3_kernel<T, Out><<<blocks_for(s.n), 256, 0, h->stream>>>(a + first * s.n_pad, a + (first + 1) * s.n_pad, a + (first + 2) * s.n_pad,
                                                                      s.id[s.cur], s.n, b, m, stage, from, stride);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host + 3 * b, stage, m * 3 * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "particle read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T>
int download_cells(fpic_handle* h, const Species& s, int32_t* cells, size_t from = 0, size_t count = ~size_t(0), size_t stride = 1)
{
    State* st = h->es;
    if (count == ~size_t(0)) count = s.n;
    if (!count) return FPIC_OK;
    const size_t chunk = 16u << 20;
    int32_t* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), std::min(chunk, count) * sizeof(int32_t)));
    const T* a = static_cast<const T*>(s.slab[s.cur]);
    for (size_t b = 0; b < count; b += chunk) {
        const size_t m = std::min(chunk, count - b);
        cells3_kernel<T><<<blocks_for(s.n), 256, 0, h->stream>>>(a, a + s.n_pad, a + 2 * s.n_pad, s.id[s.cur], s.n, b, m, st->nx, st->ny, st->nz, stage, from, stride);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(cells + b, stage, m * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { (void)hipFree(stage); return fail(h, FPIC_ERR_HIP, "cell read-back failed: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(h, hipFree(stage));
    return FPIC_OK;
}

template <typename T, typename In>
int upload_field(fpic_handle* h, const In* host, void* target)
{
    State* st = h->es;
    In* stage = nullptr;
    const size_t bytes = st->nodes * 3 * sizeof(In);
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), bytes));
    hipError_t e = hipMemcpyAsync(stage, host, bytes, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        pack_field3_kernel<T, In><<<blocks_for(st->nodes), 256, 0, h->stream>>>(stage, st->nx, st->ny, st->nz, static_cast<T*>(target), held_of(st));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "field upload failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

template <typename T, typename Out>
int download_grid(fpic_handle* h, const void* dev, size_t count, Out* host)
{
    Out* stage = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&stage), count * sizeof(Out)));
    convert_kernel<Out, T><<<blocks_for(count), 256, 0, h->stream>>>(static_cast<const T*>(dev), stage, count);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host, stage, count * sizeof(Out), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(h, FPIC_ERR_HIP, "grid read-back failed: %s", hipGetErrorString(e));
    return FPIC_OK;
}

template <typename K>
hipError_t set_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

template <typename T>
int create_state(fpic_handle* h)
{
    State* st = h->es;
    uint64_t* acc = &h->bytes_grid;
    int rc;
    if ((rc = dev_alloc(h, reinterpret_cast<void**>(&st->rho_fixed), st->nodes * sizeof(long long), acc)) ||
        (rc = dev_alloc(h, &st->rho, st->nodes * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->phi, st->nodes * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->E4, st->nodes * 4 * sizeof(T), acc)) ||
        (rc = dev_alloc(h, &st->hat, hat_values<T>(st) * 2 * sizeof(T), acc)) ||
        (rc = dev_alloc(h, reinterpret_cast<void**>(&st->spilled), sizeof(unsigned long long), acc)))
        return rc;
    const int dims[3] = { st->nx, st->ny, st->nz };
    const double len[3] = { st->lx, st->ly, st->lz };
    for (int a = 0; a < 3; ++a) {
        std::vector<double> t(dims[a]);
        const double d = len[a] / dims[a];
        for (int l = 0; l < dims[a]; ++l) {
            const double s = 2.0 / d * std::sin(kPi * l / dims[a]); // es3d_k2_table
            t[l] = s * s;
        }
        if ((rc = dev_alloc(h, reinterpret_cast<void**>(&st->k2[a]), sizeof(double) * dims[a], acc))) return rc;
        HIP_TRY(h, hipMemcpyAsync(st->k2[a], t.data(), sizeof(double) * dims[a], hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    hipError_t e;
    if ((e = hipHostMalloc(reinterpret_cast<void**>(&st->spilled_host), 2 * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipEventCreateWithFlags(&st->spill_event[0], hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&st->spill_event[1], hipEventDisableTiming)) != hipSuccess)
        return fail(h, FPIC_ERR_HIP, "handle setup failed: %s", hipGetErrorString(e));
    st->spilled_host[0] = st->spilled_host[1] = 0;
    if ((e = set_lds(push3_tiles_kernel<T, false, false>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, true, false>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, false, true>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, false, false, true>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(push3_tiles_kernel<T, true, false, true>, push3_lds_bytes<T>())) != hipSuccess ||
        (e = set_lds(bin3_count_kernel<T>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(bin3_scatter_kernel<T>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(bin3_count_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(bin3_scatter_kernel<T, EmWin<T>::LX, EmWin<T>::LY, EmWin<T>::LZ>, static_cast<size_t>(kMaxTiles3) * 4)) != hipSuccess ||
        (e = set_lds(em_push_tiles_kernel<T>, em_lds_bytes<T>())) != hipSuccess)
        return fail(h, FPIC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));

    if (st->solver == FPIC_SOLVER_YEE) {
        if ((rc = dev_alloc(h, &st->Ey, st->nodes * 4 * sizeof(T), acc)) || (rc = dev_alloc(h, &st->By, st->nodes * 4 * sizeof(T), acc)) ||
            (rc = dev_alloc(h, &st->B4n, st->nodes * 4 * sizeof(T), acc)) ||
            (rc = dev_alloc(h, reinterpret_cast<void**>(&st->Jfix), st->nodes * 3 * sizeof(long long), acc)))
            return rc;
    }
    // power-of-two grids (8 .. 512 nodes per axis): the library's own FFT passes; FPIC_POISSON_FFT=rocfft keeps rocFFT (a
    // development switch: the two agree within the solve's tolerance, tests/test_gpu_es3d.py)
    {
        const char* force = std::getenv("FPIC_POISSON_FFT");
        st->own_fft = fft_supported(st->nx) && fft_supported(st->ny) && fft_supported(st->nz) && !(force && std::strcmp(force, "rocfft") == 0);
        if (st->own_fft) {
            const int dims3[3] = { st->nx, st->ny, st->nz };
            for (int a = 0; a < 3; ++a) {
                if ((rc = dev_alloc(h, &st->fft_tw[a], static_cast<size_t>(dims3[a]) * 2 * sizeof(T), acc))) return rc;
                fft_twiddle_table_kernel<T><<<blocks_for(dims3[a]), 256, 0, h->stream>>>(static_cast<T*>(st->fft_tw[a]), dims3[a]);
            }
            HIP_TRY(h, hipGetLastError());
        }
        const size_t most = fft_lds_bytes<T>(1 << kFftMaxLog, fft_tile_columns<T>());
        // (fft_columns() launches the half-width instantiation for 512-point float columns: every form it can launch gets
        // its limit, whatever a later retune of the tile widths makes of their sizes)
        constexpr int CH = fft_tile_columns<T>() / 2;
        if ((e = set_lds(fft_x_forward_kernel<T>, most)) != hipSuccess || (e = set_lds(fft_x_inverse_kernel<T>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 0>, most)) != hipSuccess || (e = set_lds(fft_columns_kernel<T, 1>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 2>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 0, CH>, most)) != hipSuccess || (e = set_lds(fft_columns_kernel<T, 1, CH>, most)) != hipSuccess ||
            (e = set_lds(fft_columns_kernel<T, 2, CH>, most)) != hipSuccess)
            return fail(h, FPIC_ERR_HIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    if ((st->solver == FPIC_SOLVER_POISSON_FFT || st->solver == FPIC_SOLVER_YEE) && !st->own_fft) { // (YEE: the initial field is the Poisson field)
        const fdyn::RocFFT& ff = fdyn::rocfft();
        if (!ff.ok) return fail(h, FPIC_ERR_STATE, ".solver <- rocFFT is not available (%s); there is no other Poisson solver and no CPU fallback", ff.why.c_str());
        const size_t lengths[3] = { static_cast<size_t>(st->nx), static_cast<size_t>(st->ny), static_cast<size_t>(st->nz) };
        const rocfft_precision prec = sizeof(T) == 4 ? rocfft_precision_single : rocfft_precision_double;
        if ((rc = fft_status(h, ff.plan_create(&st->fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec, 3, lengths, 1, nullptr), "rocfft_plan_create (forward)")) ||
            (rc = fft_status(h, ff.plan_create(&st->inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec, 3, lengths, 1, nullptr), "rocfft_plan_create (inverse)")) ||
            (rc = fft_status(h, ff.execution_info_create(&st->info_f), "rocfft_execution_info_create")) ||
            (rc = fft_status(h, ff.execution_info_create(&st->info_i), "rocfft_execution_info_create")))
            return rc;
        size_t wf = 0, wi = 0;
        if ((rc = fft_status(h, ff.plan_get_work_buffer_size(st->fwd, &wf), "rocfft_plan_get_work_buffer_size")) ||
            (rc = fft_status(h, ff.plan_get_work_buffer_size(st->inv, &wi), "rocfft_plan_get_work_buffer_size")))
            return rc;
        if (wf) {
            if ((rc = dev_alloc(h, &st->work_f, wf, acc))) return rc;
            if ((rc = fft_status(h, ff.execution_info_set_work_buffer(st->info_f, st->work_f, wf), "rocfft_execution_info_set_work_buffer"))) return rc;
        }
        if (wi) {
            if ((rc = dev_alloc(h, &st->work_i, wi, acc))) return rc;
            if ((rc = fft_status(h, ff.execution_info_set_work_buffer(st->info_i, st->work_i, wi), "rocfft_execution_info_set_work_buffer"))) return rc;
        }
    }
    if ((rc = alloc_species<T>(h, st->sp[0]))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FPIC_OK;
}

int check_species(fpic_handle* h, int species)
{
    if (species < 0 || species >= static_cast<int>(h->es->sp.size()))
        return fail(h, FPIC_ERR_INVALID_ARG, ".species <- %d is not one of the handle's %zu species", species, h->es->sp.size());
    return FPIC_OK;
}
