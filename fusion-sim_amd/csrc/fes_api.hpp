// fes_api.hpp — the CART3D electrostatic extension behind the C ABI (fes_api.hip); the entry
// points of fpic_api.hip forward here when spec.geometry == FPIC_GEOM_CART3D.  Not part of the ABI.
#pragma once

#include "fpic_handle.hpp"

namespace fes {

int create(fpic_handle* h);
void release(fpic_handle* h);
int add_species(fpic_handle* h, double mass, double charge, uint64_t count, int* index);
int set_particles(fpic_handle* h, int species, const void* pos_aos, const void* vel_aos, uint64_t first, uint64_t n, int dtype);
uint64_t species_count(const fpic_handle* h, int species);
int get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, int dtype, uint64_t first = 0, uint64_t n = ~0ull, uint64_t stride = 1);
int get_cells(fpic_handle* h, int species, int32_t* cells, uint64_t first = 0, uint64_t n = ~0ull, uint64_t stride = 1);
int add_b(fpic_handle* h, double bx, double by, double bz);
int set_field3(fpic_handle* h, int which, const void* data, int nx, int ny, int nz, int dtype);
int read_field3(fpic_handle* h, int which, void* out, int dtype);
int precalc(fpic_handle* h);
int density(fpic_handle* h);
int substeps(fpic_handle* h, int nsub); // (step(n) = substeps(2 n))
int sort(fpic_handle* h);
int device_buffer(fpic_handle* h, int which, void** dptr, size_t* bytes);
int save_checkpoint(fpic_handle* h, const char* path);
int load_checkpoint(fpic_handle* h, const char* path);
// z-slab decomposition
int domain_init(fpic_handle* h, int rank, int world, int ghost_planes, int migrate_every, int distributed_solve);
int domain_set_particles(fpic_handle* h, int species, uint64_t n, const void* pos_aos, const void* vel_aos, uint32_t first_id, int dtype);
int domain_get_particles(fpic_handle* h, int species, void* pos_aos, void* vel_aos, uint32_t* ids, uint64_t capacity, uint64_t* n_out, int dtype);
int domain_stats(fpic_handle* h, uint64_t* migrated, uint64_t* lost);
bool is_decomposed(const fpic_handle* h);
int group_run(fpic_handle** hs, int n, int what /* 0 precalc, 1 step, 2 density */, int ncalls);
uint64_t particle_count(const fpic_handle* h);
uint64_t last_spill(const fpic_handle* h); // out-of-window deposits of the sub-step before last (lagged read-back)

} // namespace fes
