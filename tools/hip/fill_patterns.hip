// fill_patterns.hip -- diagnostic: how fast can a 1 B/voxel plane be filled, by pattern?
//   hipcc -O3 --offload-arch=gfx950 -o fill_patterns fill_patterns.hip && ./fill_patterns 1024
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) one workgroup per 64x8x8 tile, one 16-byte store per thread (the classify kernel's fill branch)
__global__ __launch_bounds__(256) void fill_tiles(uint8_t *s, int X, int Y, int tilesX, int tilesY) {
    const int tx = blockIdx.x % tilesX, ty = (blockIdx.x / tilesX) % tilesY, tz = blockIdx.x / (tilesX * tilesY);
    const int yy = ty * 8 + ((threadIdx.x >> 2) & 7), zz = tz * 8 + (threadIdx.x >> 5);
    *reinterpret_cast<uint4 *>(s + ((size_t)zz * Y + yy) * X + tx * 64 + 16 * (threadIdx.x & 3)) = make_uint4(2, 2, 2, 2);
}
// (b) the same tiles, block index mapped so that x-neighbours are adjacent blocks b, b+8 (same XCD)
__global__ __launch_bounds__(256) void fill_tiles_xcd(uint8_t *s, int X, int Y, int tilesX, int tilesY) {
    const unsigned k = blockIdx.x >> 3;
    const unsigned trow = (k / tilesX) * 8u + (blockIdx.x & 7u);
    const int tx = k % tilesX, ty = trow % tilesY, tz = trow / tilesY;
    const int yy = ty * 8 + ((threadIdx.x >> 2) & 7), zz = tz * 8 + (threadIdx.x >> 5);
    *reinterpret_cast<uint4 *>(s + ((size_t)zz * Y + yy) * X + tx * 64 + 16 * (threadIdx.x & 3)) = make_uint4(2, 2, 2, 2);
}
// (c) contiguous grid-stride, 16 bytes per thread
__global__ __launch_bounds__(256) void fill_linear(uint4 *s, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) s[i] = make_uint4(2, 2, 2, 2);
}
// (d) one workgroup per tile ROW (all tiles along x at one (ty,tz)): whole voxel rows
__global__ __launch_bounds__(256) void fill_rows(uint8_t *s, int X, int Y) {
    const int ty = blockIdx.x % (Y / 8), tz = blockIdx.x / (Y / 8);
    const int tpr = X / 16 < 256 ? X / 16 : 256, rstep = 256 / tpr, tr = threadIdx.x / tpr;
    if (tr >= rstep) return;
    for (int x = 16 * (threadIdx.x % tpr); x < X; x += 16 * tpr)
        for (int r = tr; r < 64; r += rstep)
            *reinterpret_cast<uint4 *>(s + ((size_t)(tz * 8 + (r >> 3)) * Y + ty * 8 + (r & 7)) * X + x) = make_uint4(2, 2, 2, 2);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t n = (size_t)N * N * N;
    uint8_t *d;
    CK(hipMalloc(&d, n));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int tiles = N / 64, trows = N / 8;
    auto timeit = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 8; ++r) {
            hipEventRecord(a);
            launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-28s %8.1f us  %6.2f TB/s\n", name, best * 1e3, n / best / 1e9);
    };
    timeit("tiles, 1 WG per tile", [&] { hipLaunchKernelGGL(fill_tiles, dim3(tiles * trows * trows), dim3(256), 0, 0, d, N, N, tiles, trows); });
    timeit("tiles, XCD row map", [&] { hipLaunchKernelGGL(fill_tiles_xcd, dim3(tiles * trows * trows), dim3(256), 0, 0, d, N, N, tiles, trows); });
    timeit("linear grid-stride 4096 WGs", [&] { hipLaunchKernelGGL(fill_linear, dim3(4096), dim3(256), 0, 0, (uint4 *)d, n / 16); });
    timeit("linear 1 store per thread", [&] { hipLaunchKernelGGL(fill_linear, dim3((unsigned)(n / 16 / 256)), dim3(256), 0, 0, (uint4 *)d, n / 16); });
    timeit("tile rows, 1 WG per row", [&] { hipLaunchKernelGGL(fill_rows, dim3(trows * trows), dim3(256), 0, 0, d, N, N); });
    CK(hipMemsetAsync(d, 2, n, 0));
    timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 2, n, 0); });
    return 0;
}
