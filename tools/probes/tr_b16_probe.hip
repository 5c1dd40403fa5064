// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = index; every lane passes an address; print what each lane gets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: lane i of a 16-lane group points at row (i/4), col (i%4)*4 of a [4][16] block (row pitch 16 elements); groups are 64 elements apart
    // mode 1: all lanes of a group pass the group's base address
    // mode 2: row pitch 128 elements (256 B): lane i -> row (i/4)*128 + (i%4)*4, groups at +16 columns
    int elem;
    if (mode == 0) elem = (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4;
    else if (mode == 1) elem = (l >> 4) * 64;
    else elem = (l >> 4) * 16 + ((l & 15) >> 2) * 128 + (l & 3) * 4;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds) + elem * 2;
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { if (l % 16 == 0 || l < 20) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
    }
    return 0;
}
