// probe of ds_read_b64_tr_b16 on an LDS image of 72-byte position records ([pos][3 pieces][12 ch] bf16):
// lane 4q+p of a 16-lane group supplies the address of row q (a position), columns 4p..4p+3; expects lane i to receive
// column (channel) i of the four rows.  Prints mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 36 + 64];
    for (int e = threadIdx.x; e < 64 * 36 + 64; e += 64) img[e] = (unsigned short)e;       // value = its own index
    __syncthreads();
    const int l = threadIdx.x, kg = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const int pos = 8 * kg + q;                              // rows = positions 8kg .. 8kg+3
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short *)img + pos * 72 + 8 * p;
    u32x2 r0, r1;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r0) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:24" : "=v"(r1) : "v"(addr) : "memory");     // piece 1 of the same rows
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1)::"memory");
    out[l * 4 + 0] = r0.x; out[l * 4 + 1] = r0.y; out[l * 4 + 2] = r1.x; out[l * 4 + 3] = r1.y;
}
int main() {
    unsigned *d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<unsigned> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int kg = l >> 4, i = l & 15;
        for (int piece = 0; piece < 2; ++piece)
            for (int q = 0; q < 4; ++q) {
                const unsigned got = (h[l * 4 + piece * 2 + (q >> 1)] >> (16 * (q & 1))) & 0xffff;
                const unsigned want = (8 * kg + q) * 36 + piece * 12 + i;       // element index of (pos, piece, column i)
                if (got != want) { if (bad < 8) printf("lane %d piece %d q %d: got %u want %u\n", l, piece, q, got, want); ++bad; }
            }
    }
    printf("tr_b16 probe: %d mismatches\n", bad);
    return 0;
}
