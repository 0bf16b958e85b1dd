// fes_fft.hpp — the hand-written FFT of the CART3D Poisson solve (power-of-two grids): Stockham radix-8/4/2 passes on
// tiles held in LDS, one HBM sweep per pass of the 3-D transform and ONE sweep for the whole z direction (forward
// transform, k-space factor, inverse transform fused).  No reference counterpart (the reference has no field solve in its
// step loop, empic.js:1436-1505): PARITY UNPINNED, the definition is oracle/es3d_oracle_impl.h
// (es3d_poisson: phi_hat = rho_hat / (eps0 K^2), K^2 the eigenvalues of the 3-point Laplacian, mean mode 0), which these
// kernels meet within the solve's tolerance (2e-5 fp32 / 1e-10 fp64), not bit for bit — exactly as rocFFT did.
//
// Why not rocFFT here: its strided-column kernels run a 512-point pass at 1.7 TB/s on this part
// (profiles/r02_c4_kernel_stats.csv: fft_rtc_*_len512_*_sbcc, 610 us for 1.07 GB), and the conversion, k-space and
// gradient sweeps around it cannot be fused into its kernels.  Grids that are not powers of two keep the rocFFT path.
//
// The arithmetic core (radix butterflies, the Stockham index maps) is plain C++ that also compiles for the host:
// tests/test_fft_core.py builds it with g++ and checks every supported length against a direct DFT.
#pragma once

#if defined(__HIPCC__)
#define FESFFT_HD __host__ __device__ __forceinline__
#else
#define FESFFT_HD inline
#endif

namespace fesfft {

template <typename T>
struct C2 {
    T x, y;
};

// Where point idx of a transform sits in its column: one complex of padding after every eight.  A radix-8 pass stores
// with a stride of 8 (first pass) or in runs of 8 at a stride of 64 (second pass); on 32 four-byte LDS banks those strides
// land every lane of a wavefront on two or four banks, with the padding they spread over all of them (the stride of 8
// becomes 9 complex = 18 words, which visits every even bank once in 16 lanes).  Loads are unit stride either way.
FESFFT_HD int swz(int idx) { return idx + (idx >> 3); }
FESFFT_HD int swz_len(int n) { return n + (n >> 3); }   // slots of a column of n points

// load / store through a pointer in any address space (a struct cannot be assigned across address spaces as a whole)
template <typename T, typename P> FESFFT_HD C2<T> ldc(P p, int i) { return C2<T>{ p[i].x, p[i].y }; }
template <typename T, typename P> FESFFT_HD void stc(P p, int i, C2<T> v) { p[i].x = v.x; p[i].y = v.y; }

template <typename T> FESFFT_HD C2<T> operator+(C2<T> a, C2<T> b) { return { a.x + b.x, a.y + b.y }; }
template <typename T> FESFFT_HD C2<T> operator-(C2<T> a, C2<T> b) { return { a.x - b.x, a.y - b.y }; }
template <typename T> FESFFT_HD C2<T> cmul(C2<T> a, C2<T> b) { return { a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x }; }
template <typename T> FESFFT_HD C2<T> cconj(C2<T> a) { return { a.x, -a.y }; }
// times -i (forward) / +i (inverse): the quarter turn of the transform's direction
template <typename T, bool INV> FESFFT_HD C2<T> quarter(C2<T> a) { return INV ? C2<T>{ -a.y, a.x } : C2<T>{ a.y, -a.x }; }

template <typename T, bool INV>
FESFFT_HD void dft2(C2<T>& a, C2<T>& b)
{
    const C2<T> t = a;
    a = t + b;
    b = t - b;
}

// natural order in, natural order out
template <typename T, bool INV>
FESFFT_HD void dft4(C2<T>& v0, C2<T>& v1, C2<T>& v2, C2<T>& v3)
{
    const C2<T> a = v0 + v2, b = v0 - v2, c = v1 + v3, d = quarter<T, INV>(v1 - v3);
    v0 = a + c; v1 = b + d; v2 = a - c; v3 = b - d;
}

template <typename T, bool INV>
FESFFT_HD void dft8(C2<T> (&v)[8])
{
    // even and odd halves (decimation in time), then X[k] = E[k] + W8^k O[k], X[k+4] = E[k] - W8^k O[k]
    dft4<T, INV>(v[0], v[2], v[4], v[6]);
    dft4<T, INV>(v[1], v[3], v[5], v[7]);
    const T h = static_cast<T>(0.70710678118654752440);
    const C2<T> o0 = v[1];
    const C2<T> o1 = INV ? C2<T>{ (v[3].x - v[3].y) * h, (v[3].x + v[3].y) * h } : C2<T>{ (v[3].x + v[3].y) * h, (v[3].y - v[3].x) * h };   // W8^1 = (1 -+ i) / sqrt 2
    const C2<T> o2 = quarter<T, INV>(v[5]);                                                                                                  // W8^2 = -+ i
    const C2<T> o3 = INV ? C2<T>{ (-v[7].x - v[7].y) * h, (v[7].x - v[7].y) * h } : C2<T>{ (v[7].y - v[7].x) * h, (-v[7].x - v[7].y) * h };  // W8^3 = (-1 -+ i) / sqrt 2
    const C2<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    v[0] = e0 + o0; v[4] = e0 - o0;
    v[1] = e1 + o1; v[5] = e1 - o1;
    v[2] = e2 + o2; v[6] = e2 - o2;
    v[3] = e3 + o3; v[7] = e3 - o3;
}

template <typename T, int R, bool INV>
FESFFT_HD void dft(C2<T> (&v)[R])
{
    if constexpr (R == 2) dft2<T, INV>(v[0], v[1]);
    else if constexpr (R == 4) dft4<T, INV>(v[0], v[1], v[2], v[3]);
    else dft8<T, INV>(v);
}

// One butterfly of a Stockham pass of radix R on one column (N points at col[0 .. N), unit stride), Ns = the product
// of the radices of the passes before it: butterfly j of N / R reads col[j + r N / R], turns input r by
// W_N^(r k N / (Ns R)) with k = j mod Ns (tw[t] = exp(-2 pi i t / N), conjugated for the inverse), transforms, and — after
// every butterfly of the column has read — writes col[(j div Ns) Ns R + k + r Ns].  The passes in any order of radices
// whose product is N leave the transform in natural order (autosort; unnormalised in both directions).
// (Col / Tw: pointers to C2<T> in whatever address space the caller keeps the column and the table: LDS on the device)
// (point idx of the column sits at col[idx * stride])
template <typename T, int R, bool INV, typename Col, typename Tw>
FESFFT_HD void pass_load(Col col, Tw tw, int N, int Ns, int j, C2<T> (&v)[R], int stride = 1)
{
    const int per = N / R, k = j & (Ns - 1);
    const int tstep = k * (per >> __builtin_ctz(static_cast<unsigned>(Ns)));   // N / (Ns R): Ns is a power of two (no division by a run-time number)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        v[r] = ldc<T>(col, swz(j + r * per) * stride);
        if (r && Ns > 1) {
            const C2<T> w = ldc<T>(tw, r * tstep);
            v[r] = cmul(v[r], INV ? cconj(w) : w);
        }
    }
    dft<T, R, INV>(v);
}

template <typename T, int R, typename Col>
FESFFT_HD void pass_store(Col col, int Ns, int j, const C2<T> (&v)[R], int stride = 1)
{
    const int k = j & (Ns - 1);
    const int j0 = (j - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) stc<T>(col, swz(j0 + r * Ns) * stride, v[r]);
}

// the radix of the next pass when 2^rem points are still to be factored: 8 while it divides, then 4 or 2
FESFFT_HD int next_radix_log(int rem) { return rem >= 3 ? 3 : rem; }

} // namespace fesfft

#if defined(__HIPCC__)

#include "fpic_kernels.hpp"

namespace fes {

using fesfft::C2;

#if !defined(FES_FFT_VEC)
#define FES_FFT_VEC 1
#endif
constexpr int kFftThreads = 512;
constexpr int kFftMaxLog = 9, kFftMinLog = 3;           // 8 .. 512 points per axis (a padded tile of 1024-point double columns would not fit the LDS; such grids keep rocFFT)
// columns of a tile: 128 contiguous bytes of a row (16 complex floats, 8 complex doubles)
template <typename T> constexpr int fft_tile_columns() { return static_cast<int>(128 / sizeof(C2<T>)); }
// LDS of a tile of `cols` transforms of N points — each column N + N / 8 slots (fesfft::swz) and one more complex of
// padding, so that consecutive columns start two banks apart and the transposing loads / stores of a tile are free of
// bank conflicts too — + the twiddle table
template <typename T> constexpr size_t fft_lds_bytes(int N, int cols) { return (static_cast<size_t>(cols) * (N + (N >> 3) + 1) + N) * sizeof(C2<T>); }

inline bool fft_supported(int n) { return n >= (1 << kFftMinLog) && n <= (1 << kFftMaxLog) && (n & (n - 1)) == 0; }
inline int fft_log2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

template <typename T> __device__ __forceinline__ void sincospi_(T a, T* s, T* c);
template <> __device__ __forceinline__ void sincospi_<float>(float a, float* s, float* c) { sincospif(a, s, c); }
template <> __device__ __forceinline__ void sincospi_<double>(double a, double* s, double* c) { sincospi(a, s, c); }

// tw[t] = exp(-2 pi i t / N), formed in double and rounded once: a table per axis in global memory, made once per handle
// (forming it per workgroup cost a quarter of a column pass at 256 points) ...
template <typename T>
__global__ __launch_bounds__(256) void fft_twiddle_table_kernel(T* __restrict__ table, int N)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    double s, c;
    sincospi(-2.0 * static_cast<double>(t) / static_cast<double>(N), &s, &c);
    table[2 * t] = static_cast<T>(c);
    table[2 * t + 1] = static_cast<T>(s);
}
// ... and copied into LDS by every workgroup
template <typename T>
__device__ __forceinline__ void fft_twiddles(FPIC_LDS C2<T>* tw, const T* __restrict__ table, int N)
{
    const C2<T>* g = reinterpret_cast<const C2<T>*>(table);
    for (int t = threadIdx.x; t < N; t += kFftThreads) fesfft::stc<T>(tw, t, g[t]);
}

// one pass of radix R over `cols` columns of N points in LDS (column c at buf + c * ld), in place: a round of the loop
// takes whole columns (N / R butterflies each, a power of two that divides the workgroup), so "every butterfly of the
// column has read" is one barrier
template <typename T, int R, bool INV>
__device__ __forceinline__ void fft_pass(FPIC_LDS C2<T>* buf, int ld, int cols, const FPIC_LDS C2<T>* tw, int N, int Ns, int logn)
{
    constexpr int kLogR = R == 8 ? 3 : (R == 4 ? 2 : 1);
    const int per = N >> kLogR;                  // <= 512 = kFftThreads for N <= 1024, R >= 2; a power of two
    const int total = per * cols;
    for (int b0 = 0; b0 < total; b0 += kFftThreads) {
        const int b = b0 + static_cast<int>(threadIdx.x);
        const bool active = b < total;
        const int c = b >> (logn - kLogR), j = b & (per - 1);
        C2<T> v[R];
        FPIC_LDS C2<T>* col = buf + c * ld;
        if (active) fesfft::pass_load<T, R, INV>(col, tw, N, Ns, j, v);
        __syncthreads();
        if (active) fesfft::pass_store<T, R>(col, Ns, j, v);
        __syncthreads();
    }
}

// all passes of a transform of N = 2^logn points on the tile
template <typename T, bool INV>
__device__ __forceinline__ void fft_tile(FPIC_LDS C2<T>* buf, int ld, int cols, const FPIC_LDS C2<T>* tw, int N, int logn)
{
    int Ns = 1;
    for (int rem = logn; rem > 0;) {
        const int rl = fesfft::next_radix_log(rem);
        if (rl == 3) fft_pass<T, 8, INV>(buf, ld, cols, tw, N, Ns, logn);
        else if (rl == 2) fft_pass<T, 4, INV>(buf, ld, cols, tw, N, Ns, logn);
        else fft_pass<T, 2, INV>(buf, ld, cols, tw, N, Ns, logn);
        Ns <<= rl;
        rem -= rl;
    }
}

// ---- x pass, forward: rows of nx real values -> rows of nx / 2 + 1 complex values.  The real values are the charge
// grid itself, rho = T((double)fixed * scale) (es3d_rho_real) formed while the row is loaded: no separate conversion
// sweep.  TWO real rows a, b ride on one complex transform z = a + i b; their spectra come apart as
// A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i (exact halvings).  A workgroup takes 2 * pairs_per_wg rows.
template <typename T>
__global__ __launch_bounds__(kFftThreads) void fft_x_forward_kernel(const long long* __restrict__ fixed, const T* __restrict__ rho, double scale, size_t rows, int nx, int logn,
                                                                    int pairs_per_wg, T* __restrict__ hat, const T* __restrict__ twt, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fft_lds[];
    FPIC_LDS C2<T>* buf = (FPIC_LDS C2<T>*)fft_lds;
    const int ld = fesfft::swz_len(nx) + 1, nxh = nx / 2 + 1;
    FPIC_LDS C2<T>* tw = buf + pairs_per_wg * ld;
    fft_twiddles<T>(tw, twt, nx);
    const size_t row0 = static_cast<size_t>(blockIdx.x) * (2 * pairs_per_wg);
    const int pairs = static_cast<int>((rows - row0 < static_cast<size_t>(2 * pairs_per_wg) ? rows - row0 : 2 * pairs_per_wg) / 2); // (rows is even: ny is a power of two)
    if (fixed && FES_FFT_VEC) {
        // two accumulators — 16 bytes — per lane and load (nx is even; rows of the int64 grid are 16-byte aligned)
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        const int half_nx = nx / 2;   // (a power of two: the row and the column of an element are a shift and a mask)
        for (int e = threadIdx.x; e < pairs * half_nx; e += kFftThreads) {
            const int p = e >> (logn - 1), i = 2 * (e & (half_nx - 1));
            const size_t g = (row0 + 2 * p) * nx + i;
            // (the charge grid is read once per sub-step: non-temporal)
            const ll2 va = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(fixed + g)), vb = __builtin_nontemporal_load(reinterpret_cast<const ll2*>(fixed + g + nx));
            fesfft::stc<T>(buf, p * ld + fesfft::swz(i), C2<T>{ static_cast<T>(static_cast<double>(va.x) * scale), static_cast<T>(static_cast<double>(vb.x) * scale) });
            fesfft::stc<T>(buf, p * ld + fesfft::swz(i + 1), C2<T>{ static_cast<T>(static_cast<double>(va.y) * scale), static_cast<T>(static_cast<double>(vb.y) * scale) });
        }
    } else {
        for (int e = threadIdx.x; e < pairs * nx; e += kFftThreads) {
            const int p = e / nx, i = e - p * nx;
            const size_t g = (row0 + 2 * p) * nx + i;
            // (rho given: the charge density already converted — a rank of a replicated solve has gathered the others' planes)
            const T a = fixed ? static_cast<T>(static_cast<double>(__builtin_nontemporal_load(fixed + g)) * scale) : rho[g];
            const T b = fixed ? static_cast<T>(static_cast<double>(__builtin_nontemporal_load(fixed + g + nx)) * scale) : rho[g + nx];
            fesfft::stc<T>(buf, p * ld + fesfft::swz(i), C2<T>{ a, b });
        }
    }
    __syncthreads();
    fft_tile<T, false>(buf, ld, pairs, tw, nx, logn);
    C2<T>* out = reinterpret_cast<C2<T>*>(hat);
    const T half = static_cast<T>(0.5);
    auto spectra = [&](int p, int k, C2<T>& A, C2<T>& B) {
        const C2<T> z = fesfft::ldc<T>(buf, p * ld + fesfft::swz(k)), w = fesfft::ldc<T>(buf, p * ld + fesfft::swz((nx - k) & (nx - 1)));
        A = C2<T>{ (z.x + w.x) * half, (z.y - w.y) * half };
        B = C2<T>{ (z.y + w.y) * half, (w.x - z.x) * half };
    };
    if constexpr (sizeof(T) == 4 && FES_FFT_VEC) {
        // float: two spectral values — 16 bytes — per lane and store (rows start 16-byte aligned: the pitch is a multiple of the
        // tile width); k = nx / 2, the odd one out of nx / 2 + 1, goes alone
        using V4 = typename fpic::NatVec16<T>::type;
        const int h2 = nx / 4;   // pairs (k, k + 1), k even, k + 1 < nx / 2 + 1: a power of two per row
        for (int p = threadIdx.x; p < pairs; p += kFftThreads) {
            C2<T> A0, B0;
            spectra(p, nx / 2, A0, B0);
            out[(row0 + 2 * p) * pitch + nx / 2] = A0;
            out[(row0 + 2 * p + 1) * pitch + nx / 2] = B0;
        }
        for (int e = threadIdx.x; e < pairs * h2; e += kFftThreads) {
            const int p = e >> (logn - 2), q = e & (h2 - 1);
            C2<T> A0, B0, A1, B1;
            spectra(p, 2 * q, A0, B0);
            spectra(p, 2 * q + 1, A1, B1);
            *reinterpret_cast<V4*>(out + (row0 + 2 * p) * pitch + 2 * q) = V4{ A0.x, A0.y, A1.x, A1.y };
            *reinterpret_cast<V4*>(out + (row0 + 2 * p + 1) * pitch + 2 * q) = V4{ B0.x, B0.y, B1.x, B1.y };
        }
    } else {
        for (int e = threadIdx.x; e < pairs * nxh; e += kFftThreads) {
            const int p = e / nxh, k = e - p * nxh;
            C2<T> A, B;
            spectra(p, k, A, B);
            out[(row0 + 2 * p) * pitch + k] = A;
            out[(row0 + 2 * p + 1) * pitch + k] = B;
        }
    }
}

// ---- x pass, inverse: rows of nx / 2 + 1 complex values (the half spectra of real rows) -> rows of nx real values; two
// rows per complex transform again: Z[k] = A[k] + i B[k], Z[N-k] = conj A[k] + i conj B[k]; a = Re z, b = Im z
template <typename T>
__global__ __launch_bounds__(kFftThreads) void fft_x_inverse_kernel(const T* __restrict__ hat, size_t rows, int nx, int logn, int pairs_per_wg, T* __restrict__ phi,
                                                                    const T* __restrict__ twt, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fft_lds[];
    FPIC_LDS C2<T>* buf = (FPIC_LDS C2<T>*)fft_lds;
    const int ld = fesfft::swz_len(nx) + 1, nxh = nx / 2 + 1;
    FPIC_LDS C2<T>* tw = buf + pairs_per_wg * ld;
    fft_twiddles<T>(tw, twt, nx);
    const size_t row0 = static_cast<size_t>(blockIdx.x) * (2 * pairs_per_wg);
    const int pairs = static_cast<int>((rows - row0 < static_cast<size_t>(2 * pairs_per_wg) ? rows - row0 : 2 * pairs_per_wg) / 2);
    const C2<T>* in = reinterpret_cast<const C2<T>*>(hat);
    auto place = [&](int p, int k, C2<T> A, C2<T> B) {
        fesfft::stc<T>(buf, p * ld + fesfft::swz(k), C2<T>{ A.x - B.y, A.y + B.x });
        if (k && k < nx - k) fesfft::stc<T>(buf, p * ld + fesfft::swz(nx - k), C2<T>{ A.x + B.y, B.x - A.y });
    };
    if constexpr (sizeof(T) == 4 && FES_FFT_VEC) {
        // float: two spectral values — 16 bytes — per lane and load; k = nx / 2 goes alone (as in the forward pass's store)
        using V4 = typename fpic::NatVec16<T>::type;
        const int h2 = nx / 4;
        for (int p = threadIdx.x; p < pairs; p += kFftThreads) place(p, nx / 2, in[(row0 + 2 * p) * pitch + nx / 2], in[(row0 + 2 * p + 1) * pitch + nx / 2]);
        for (int e = threadIdx.x; e < pairs * h2; e += kFftThreads) {
            const int p = e >> (logn - 2), q = e & (h2 - 1);
            const V4 a = *reinterpret_cast<const V4*>(in + (row0 + 2 * p) * pitch + 2 * q), b = *reinterpret_cast<const V4*>(in + (row0 + 2 * p + 1) * pitch + 2 * q);
            place(p, 2 * q, C2<T>{ a.x, a.y }, C2<T>{ b.x, b.y });
            place(p, 2 * q + 1, C2<T>{ a.z, a.w }, C2<T>{ b.z, b.w });
        }
    } else {
        for (int e = threadIdx.x; e < pairs * nxh; e += kFftThreads) {
            const int p = e / nxh, k = e - p * nxh;
            place(p, k, in[(row0 + 2 * p) * pitch + k], in[(row0 + 2 * p + 1) * pitch + k]);
        }
    }
    __syncthreads();
    fft_tile<T, true>(buf, ld, pairs, tw, nx, logn);
    if constexpr (sizeof(T) == 4 && FES_FFT_VEC) {
        // float: four consecutive values of a row — 16 bytes — per lane and store (nx >= 8 is a power of two)
        using V4 = typename fpic::NatVec16<T>::type;
        const int quads = nx / 4;
        for (int e = threadIdx.x; e < pairs * quads; e += kFftThreads) {
            const int p = e >> (logn - 2), i = 4 * (e & (quads - 1));
            const C2<T> z0 = fesfft::ldc<T>(buf, p * ld + fesfft::swz(i)), z1 = fesfft::ldc<T>(buf, p * ld + fesfft::swz(i + 1)),
                        z2 = fesfft::ldc<T>(buf, p * ld + fesfft::swz(i + 2)), z3 = fesfft::ldc<T>(buf, p * ld + fesfft::swz(i + 3));
            const size_t g = (row0 + 2 * p) * nx + i;
            *reinterpret_cast<V4*>(phi + g) = V4{ z0.x, z1.x, z2.x, z3.x };
            *reinterpret_cast<V4*>(phi + g + nx) = V4{ z0.y, z1.y, z2.y, z3.y };
        }
    } else {
        for (int e = threadIdx.x; e < pairs * nx; e += kFftThreads) {
            const int p = e / nx, i = e - p * nx;
            const C2<T> z = fesfft::ldc<T>(buf, p * ld + fesfft::swz(i));
            const size_t g = (row0 + 2 * p) * nx + i;
            phi[g] = z.x;
            phi[g + nx] = z.y;
        }
    }
}

// ---- column passes.  The half spectrum is [outer2][N or outer1][...][pitch] with the x index fastest and rows of
// pitch >= nxh complex values (row_pitch in fes_api.hip: a multiple of the tile width, so that a tile's C consecutive
// values of a row are ONE aligned 128-byte line instead of pieces of two: 129 -> 144 floats, 136 doubles); a tile is C
// consecutive x of every point of ONE column line: element (idx, c) of tile (o, t) sits at
//     base + o * outer_stride + idx * stride + t * C + c.
// MODE 0: forward transform; 1: inverse; 2: forward, k-space factor, inverse — the whole z direction of the solve in one
// sweep (the factor needs the mode indices: kx = t C + c, ky = y0 + o, kz = idx).
struct ColLayout {
    size_t outer_stride, stride;   // complex elements
    int outer, nxh, pitch;
    // The y passes of a slab-decomposed solve exchange their lines with the other ranks (an all-to-all transposition): the
    // forward pass STORES straight into the send buffer [q][nzl][nyl][pitch] — row ky = q nyl + yl of plane o goes to rank q
    // — and the inverse pass LOADS from the receive buffer of the same shape (nyl = 0: plain layout on both sides).  That
    // is the pack / unpack sweep of round 2 (26 + 39 us per rank at 512^3 / 8) done by address arithmetic.
    int nyl, nzl;
};

// element (idx, c) of the tile of line o in the exchange buffer: [q][nzl][nyl][pitch]
__device__ __forceinline__ size_t exchange_index(const ColLayout& L, int o, int idx, int i)
{
    const int q = idx / L.nyl, yl = idx - q * L.nyl;
    return ((static_cast<size_t>(q) * L.nzl + o) * L.nyl + yl) * L.pitch + i;
}

template <typename T, int MODE, int C = fft_tile_columns<T>()>
__global__ __launch_bounds__(kFftThreads) void fft_columns_kernel(T* __restrict__ hat, T* __restrict__ xbuf, ColLayout L, int N, int logn, int y0,
                                                                  const double* __restrict__ k2x, const double* __restrict__ k2y, const double* __restrict__ k2z,
                                                                  double inv_eps0_n, const T* __restrict__ twt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fft_lds[];
    FPIC_LDS C2<T>* buf = (FPIC_LDS C2<T>*)fft_lds;
    const int ld = fesfft::swz_len(N) + 1;
    FPIC_LDS C2<T>* tw = buf + C * ld;
    fft_twiddles<T>(tw, twt, N);
    const int tiles = (L.nxh + C - 1) / C;
    const int o = static_cast<int>(blockIdx.x / tiles), t = static_cast<int>(blockIdx.x % tiles);
    const int i0 = t * C, cols = L.nxh - i0 < C ? L.nxh - i0 : C;
    C2<T>* base = reinterpret_cast<C2<T>*>(hat) + static_cast<size_t>(o) * L.outer_stride + i0;
    C2<T>* xb = reinterpret_cast<C2<T>*>(xbuf);
    const bool load_exchanged = MODE == 1 && L.nyl > 0, store_exchanged = MODE == 0 && L.nyl > 0;
    // (float: two complex values — 16 bytes — per lane and access; rows are padded to whole tiles, so the pair of an odd last
    // column is inside the row's padding: loaded and stored like the rest, never transformed.  FES_FFT_VEC=0: one per lane)
    constexpr bool kPairs = sizeof(T) == 4 && (C % 2) == 0 && FES_FFT_VEC;
    if constexpr (kPairs) {
        using V4 = typename fpic::NatVec16<T>::type;
        constexpr int H = C / 2;
        for (int e = threadIdx.x; e < N * H; e += kFftThreads) {
            const int idx = e / H, c = 2 * (e - idx * H);
            if (c >= cols) continue;
            const C2<T>* src = load_exchanged ? xb + exchange_index(L, o, idx, i0 + c) : base + static_cast<size_t>(idx) * L.stride + c;
            const V4 v = *reinterpret_cast<const V4*>(src);
            fesfft::stc<T>(buf, c * ld + fesfft::swz(idx), C2<T>{ v.x, v.y });
            fesfft::stc<T>(buf, (c + 1) * ld + fesfft::swz(idx), C2<T>{ v.z, v.w });
        }
    } else {
        for (int e = threadIdx.x; e < N * C; e += kFftThreads) {
            const int idx = e / C, c = e - idx * C;
            if (c < cols)
                fesfft::stc<T>(buf, c * ld + fesfft::swz(idx), load_exchanged ? xb[exchange_index(L, o, idx, i0 + c)] : base[static_cast<size_t>(idx) * L.stride + c]);
        }
    }
    __syncthreads();
    if constexpr (MODE == 1) fft_tile<T, true>(buf, ld, cols, tw, N, logn);
    else fft_tile<T, false>(buf, ld, cols, tw, N, logn);
    if constexpr (MODE == 2) {
        // phi_hat = rho_hat / (eps0 K^2 N), K^2 = (k2x + k2y) + k2z in double, the mean mode 0 (es3d_poisson, kspace_kernel)
        const int j = y0 + o;
        for (int e = threadIdx.x; e < N * cols; e += kFftThreads) {
            const int c = e >> logn, k = e & (N - 1);   // (N = 2^logn)
            const int i = i0 + c;
            const double K2 = (k2x[i] + k2y[j]) + k2z[k];
            // 1 / K2 by the hardware's reciprocal and one Newton step (an IEEE double division is ~40 instructions, and eight of
            // them per thread and tile were a quarter of this sweep's arithmetic): within 2 ulp of the quotient in double, i.e.
            // the same float in all but boundary cases — inside the solve's tolerance either way (2e-5 / 1e-10)
            double r = __builtin_amdgcn_rcp(K2);
            r = __builtin_fma(__builtin_fma(-K2, r, 1.0), r, r);
            const T g = (i | j | k) ? static_cast<T>(inv_eps0_n * r) : static_cast<T>(0);
            const C2<T> v = fesfft::ldc<T>(buf, c * ld + fesfft::swz(k));
            fesfft::stc<T>(buf, c * ld + fesfft::swz(k), C2<T>{ v.x * g, v.y * g });
        }
        __syncthreads();
        fft_tile<T, true>(buf, ld, cols, tw, N, logn);
    }
    if constexpr (kPairs) {
        using V4 = typename fpic::NatVec16<T>::type;
        constexpr int H = C / 2;
        for (int e = threadIdx.x; e < N * H; e += kFftThreads) {
            const int idx = e / H, c = 2 * (e - idx * H);
            if (c >= cols) continue;
            const C2<T> a = fesfft::ldc<T>(buf, c * ld + fesfft::swz(idx)), b = fesfft::ldc<T>(buf, (c + 1) * ld + fesfft::swz(idx));
            C2<T>* dst = store_exchanged ? xb + exchange_index(L, o, idx, i0 + c) : base + static_cast<size_t>(idx) * L.stride + c;
            *reinterpret_cast<V4*>(dst) = V4{ a.x, a.y, b.x, b.y };
        }
    } else {
        for (int e = threadIdx.x; e < N * C; e += kFftThreads) {
            const int idx = e / C, c = e - idx * C;
            if (c >= cols) continue;
            const C2<T> v = fesfft::ldc<T>(buf, c * ld + fesfft::swz(idx));
            if (store_exchanged) xb[exchange_index(L, o, idx, i0 + c)] = v;
            else base[static_cast<size_t>(idx) * L.stride + c] = v;
        }
    }
}

} // namespace fes

#endif // __HIPCC__
