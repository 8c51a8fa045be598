// probe: which SIMD does wave w of an 8-wave (and 4-wave) workgroup land on?  prints HW_ID fields for a few workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the waves resident together for a moment so that placement is the steady-state one
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(32);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = id;
        out[2 * w + 1] = xcc;
    }
}
int main() {
    for (int wpb : {8, 4}) {
        const int blocks = wpb == 8 ? 256 : 512;
        unsigned *d;
        hipMalloc(&d, blocks * wpb * 2 * sizeof(unsigned));
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(wpb * 64), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(blocks * wpb * 2);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        printf("waves per workgroup %d\n", wpb);
        for (int b : {0, 1, 2, 9, 100}) {
            printf(" wg %3d:", b);
            for (int w = 0; w < wpb; ++w) {
                const unsigned id = h[2 * (b * wpb + w)], x = h[2 * (b * wpb + w) + 1];
                printf("  [xcc%u se%u cu%2u simd%u slot%u]", x & 15, (id >> 13) & 7, (id >> 8) & 15, (id >> 4) & 3, id & 15);
            }
            printf("\n");
        }
        hipFree(d);
    }
    return 0;
}
