// ablate_lds_atomics.hip — development probe: LDS atomic add rate by operand type on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ idx, size_t n, T* out, int cells)
{
    extern __shared__ unsigned char raw[];
    T* tile = reinterpret_cast<T*>(raw);
    for (int i = threadIdx.x; i < cells * 4; i += 256) tile[i] = T(0);
    __syncthreads();
    const size_t per = 32768;
    const size_t b = blockIdx.x * per;
    for (size_t i = b + threadIdx.x; i < b + per && i < n; i += 256) {
        const unsigned c = idx[i] % cells;
        T* t = tile + 4 * c;
        atomicAdd(t, T(1)); atomicAdd(t + 1, T(2)); atomicAdd(t + 2, T(3)); atomicAdd(t + 3, T(4));
    }
    __syncthreads();
    T acc = T(0);
    for (int i = threadIdx.x; i < cells * 4; i += 256) acc += tile[i];
    if (acc == T(123457)) out[0] = acc;
}
__global__ void fill(unsigned* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    unsigned long long z = (i + 77) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
    p[i] = (unsigned)(z >> 20);
}
template <typename T> float run(const unsigned* idx, size_t n, void* out, int cells) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned nb = (unsigned)((n + 32767) / 32768);
    const size_t sh = (size_t)cells * 4 * sizeof(T);
    CK(hipFuncSetAttribute((const void*)k<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    k<T, 1><<<nb, 256, sh>>>(idx, n, (T*)out, cells); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) k<T, 1><<<nb, 256, sh>>>(idx, n, (T*)out, cells);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
}
int main() {
    const size_t n = 100000000ull; unsigned* idx; void* out;
    CK(hipMalloc((void**)&idx, n * 4)); CK(hipMalloc(&out, 64));
    fill<<<(n + 255) / 256, 256>>>(idx, n); CK(hipDeviceSynchronize());
    const int cells = 48 * 48;
    printf("4 LDS atomic adds per element, 1e8 elements, %d cells, 4 B index stream (0.4 GB)\n", cells);
    printf("float              %.3f ms\n", run<float>(idx, n, out, cells));
    printf("unsigned int       %.3f ms\n", run<unsigned>(idx, n, out, cells));
    printf("int                %.3f ms\n", run<int>(idx, n, out, cells));
    printf("unsigned long long %.3f ms\n", run<unsigned long long>(idx, n, out, cells));
    printf("double             %.3f ms\n", run<double>(idx, n, out, cells));
    return 0;
}
