// Does v_cvt_rpi_i32_f32 compute floor(x + 0.5) without rounding the sum?  Compares it with
// floor(x) + (fract(x) >= 0.5), which is std::round for x > -0.5, on EVERY float in
// (-0.5, 2^24): prints the number of mismatches and the first few.
//   hipcc --offload-arch=gfx950 -O2 -o cvt_rpi_check tools/hip/cvt_rpi_check.hip && ./cvt_rpi_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ void check(unsigned lo, unsigned hi, unsigned long long *nbad, unsigned *first) {
    const unsigned long long i = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > hi) return;
    const unsigned bits = (unsigned)i;
    float x = __uint_as_float(bits);
    int a;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(a) : "v"(x));
    const int b = (int)floorf(x) + (__builtin_amdgcn_fractf(x) >= 0.5f ? 1 : 0);
    if (a != b) {
        const unsigned long long k = atomicAdd(nbad, 1ull);
        if (k < 8) first[k] = bits;
    }
}

static void run(unsigned lo, unsigned hi, const char *what) {
    unsigned long long *nbad;
    unsigned *first;
    (void)hipMalloc(&nbad, 8);
    (void)hipMalloc(&first, 32);
    (void)hipMemset(nbad, 0, 8);
    (void)hipMemset(first, 0, 32);
    const unsigned long long n = (unsigned long long)hi - lo + 1;
    check<<<(unsigned)((n + 255) / 256), 256>>>(lo, hi, nbad, first);
    unsigned long long h;
    unsigned f[8];
    (void)hipMemcpy(&h, nbad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
    printf("%s: %llu floats, %llu mismatches", what, n, h);
    for (unsigned k = 0; k < (h < 8 ? h : 8); ++k) {
        float x;
        memcpy(&x, &f[k], 4);
        printf(" [%08x = %.9g]", f[k], x);
    }
    printf("\n");
}

int main() {
    run(0x00000000u, 0x4B800000u, "[+0, 2^24]");          // all non-negative floats up to 2^24
    run(0x80000000u, 0xBEFFFFFFu, "(-0.5, -0]");          // negative floats above -0.5
    return 0;
}
