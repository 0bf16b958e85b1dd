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
constexpr int kDepositChunk = 16384;         // particles per workgroup and chunk (sweep: profiles/r01_rebin_ablation.txt)
constexpr int kMaxTiles = 16384;             // LDS histogram limit of the binning pass

struct Constants {
    double h, factor_r, factor_z, step_factor, f_rz, f_zr;
};

Constants derive_constants(const fpic_spec& s);
double shader_literal(double x);
void build_stamp(float w[kStampCells]);

} // namespace fpic
