// Micro-benchmark (analysis tooling, not product): VALU issue rate of wave64 f32 ops on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], s, 0.5f);
            if (MODE == 1) a[i] = __builtin_amdgcn_exp2f(a[i]) * s;
            if (MODE == 2) a[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), it & 63)) + s;
            if (MODE == 3 && (i & 1) == 0) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {a[i], a[i + 1]}, m = {s, s}, c = {0.5f, 0.25f};
                v = __builtin_elementwise_fma(v, m, c);
                a[i] = v.x; a[i + 1] = v.y;
            }
            if (MODE == 4) a[i] = a[i] * s;
            if (MODE == 5) a[i] = fminf(a[i] + s, 3.0f);
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE>
void run(const char* name, int ops_per_iter) {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 0.999f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 8 * ops_per_iter;  // wave-instructions
    double per_simd_cycle = winstr / (1024.0 * ms * 1e-3 * 2.4e9);
    printf("%s: %.3f ms, %.3g wave-instr, %.3f wave-instr/SIMD/cycle@2.4GHz (cycles per instr %.2f)\n", name, ms, winstr, per_simd_cycle, 1.0 / per_simd_cycle);
}
int main() { run<0>("v_fma_f32", 1); run<1>("v_exp_f32+v_mul", 2); run<2>("v_readlane+v_add", 2);
  run<3>("v_pk_fma_f32 (counted as 0.5 instr per float)", 1); run<4>("v_mul_f32", 1); run<5>("v_add+v_min", 2); return 0; }
