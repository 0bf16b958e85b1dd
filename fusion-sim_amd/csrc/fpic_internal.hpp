// fpic_internal.hpp — shared declarations of libfusionpic.so (not part of the ABI).
#pragma once

#include "../../include/fusionpic.h"

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace fpic {

constexpr int kStampSide = 11;               // empic.js:949
constexpr int kStampCells = kStampSide * kStampSide;
constexpr int kStampReach = 5;               // (nshape-1)/2
constexpr int kCdfSide = 512;                // empic.js:228-241
constexpr int kEntropySide = 1024;           // empic.js:142

// Cell tiles used to bin particles for the LDS-staged scatter.
constexpr int kTileSide = 32;                // cells per tile edge
constexpr int kTileHalo = 8;                 // extra cells kept in LDS around a tile
constexpr int kTileLds = kTileSide + 2 * kTileHalo;
#if !defined(FPIC_DEPOSIT_CHUNK)
#define FPIC_DEPOSIT_CHUNK 16384
#endif
constexpr int kDepositChunk = FPIC_DEPOSIT_CHUNK; // particles per workgroup and chunk (sweep: profiles/r01_rebin_ablation.txt)
constexpr int kMaxTiles = 16384;             // LDS histogram limit of the binning pass

// The per-cell sums of density()'s point sprites live on the cells a sprite's CENTRE can have: columns 0..nr (r = 1
// lands on column nr) under the ideal convention, and up to 5 cells outside the grid when the sprites are drawn as a
// rasteriser draws them (spec.raster_subpixel_bits: a point whose centre has left the target is cropped, not dropped).
// Grid of (nr + 1 + 2*kSumsApron) x (nz + 1 + 2*kSumsApron) cells, cell (ic, jc) at sums_index(ic, jc, nr).
constexpr int kSumsApron = 5;
constexpr size_t sums_width(int nr) { return static_cast<size_t>(nr) + 1 + 2 * kSumsApron; }
constexpr size_t sums_cells(int nr, int nz) { return sums_width(nr) * sums_width(nz); }
constexpr size_t sums_index(int ic, int jc, int nr) { return static_cast<size_t>(ic + kSumsApron) + sums_width(nr) * static_cast<size_t>(jc + kSumsApron); }
constexpr bool sums_holds(int ic, int jc, int nr, int nz) { return ic >= -kSumsApron && ic <= nr + kSumsApron && jc >= -kSumsApron && jc <= nz + kSumsApron; }

struct Constants {
    double h, factor_r, factor_z, step_factor, f_rz, f_zr;
};

Constants derive_constants(const fpic_spec& s);
double shader_literal(double x);
void build_stamp(float w[kStampCells]);

} // namespace fpic
