#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_xcc(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x * 2] = x;
        out[blockIdx.x * 2 + 1] = hw;
    }
}
int main() {
    for (int T : {256, 1024}) {
        const int G = 512;
        unsigned* d;
        hipMalloc(&d, G * 8);
        hipMemset(d, 0xFF, G * 8);
        hipLaunchKernelGGL(k_xcc, dim3(G), dim3(T), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(G * 2);
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        printf("threads %d: xcc_id (raw & 0xF) of workgroups 0..63:\n", T);
        for (int i = 0; i < 64; ++i) printf("%u%s", h[i * 2] & 0xF, (i % 16 == 15) ? "\n" : " ");
        printf("raw[0..7]: ");
        for (int i = 0; i < 8; ++i) printf("%08x ", h[i * 2]);
        printf("\n");
        int same_res = 0;
        for (int i = 8; i < G; ++i) same_res += ((h[i * 2] & 0xF) == (h[(i % 8) * 2] & 0xF));
        printf("workgroups b >= 8 on the XCC of workgroup b %% 8: %d of %d\n", same_res, G - 8);
        hipFree(d);
    }
    return 0;
}
