// Issue cost of the vector instructions the exact carve kernel is made of, one wave on one
// SIMD, dependent chains of 8 independent streams: cycles per instruction from s_memtime.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rates tools/hip/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 512
template <int OP>
__global__ void rate_kernel(unsigned long long *out, double seed) {
    double d[8];
    float f[8];
    for (int i = 0; i < 8; ++i) {
        d[i] = seed + threadIdx.x * 1e-3 + i;
        f[i] = (float)d[i];
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) d[i] = d[i] + 1.0000001;                                   // v_add_f64
            if (OP == 1) d[i] = __builtin_fma(d[i], 1.0000001, 0.5);                // v_fma_f64
            if (OP == 2) d[i] = d[i] * 1.0000001;                                   // v_mul_f64
            if (OP == 3) { f[i] = (float)d[i]; d[i] = d[i] + (double)f[i]; }        // cvt both ways + add
            if (OP == 4) f[i] = __builtin_fmaf(f[i], 1.0000001f, 0.5f);             // v_fma_f32
            if (OP == 5) f[i] = __builtin_amdgcn_rcpf(f[i]) + 1.0f;                 // v_rcp_f32 + add
            if (OP == 6) f[i] = __builtin_rintf(f[i] * 1.3f);                       // mul + rndne
            if (OP == 7) f[i] = __builtin_floorf(f[i]) + __builtin_amdgcn_fractf(f[i] * 1.1f);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double acc = 0;
    for (int i = 0; i < 8; ++i) acc += d[i] + f[i];
    if (threadIdx.x == 0) {
        out[2 * OP] = t1 - t0;
        out[2 * OP + 1] = (unsigned long long)acc;
    }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 16 * sizeof(unsigned long long));
    rate_kernel<0><<<1, 64>>>(d, 1.0);
    rate_kernel<1><<<1, 64>>>(d, 1.0);
    rate_kernel<2><<<1, 64>>>(d, 1.0);
    rate_kernel<3><<<1, 64>>>(d, 1.0);
    rate_kernel<4><<<1, 64>>>(d, 1.0);
    rate_kernel<5><<<1, 64>>>(d, 1.0);
    rate_kernel<6><<<1, 64>>>(d, 1.0);
    rate_kernel<7><<<1, 64>>>(d, 1.0);
    std::vector<unsigned long long> h(16);
    hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
    const char *names[8] = {"v_add_f64", "v_fma_f64", "v_mul_f64", "cvt_f32_f64 + cvt_f64_f32 + add_f64",
                            "v_fma_f32", "v_rcp_f32 + v_add_f32", "v_mul_f32 + v_rndne_f32",
                            "v_floor + v_mul + v_fract + v_add"};
    for (int i = 0; i < 8; ++i)
        printf("%-40s %7.2f ticks per group of ops (x8 streams, %d reps; s_memtime ticks)\n", names[i],
               (double)h[2 * i] / (REP * 8.0), REP);
    return 0;
}
