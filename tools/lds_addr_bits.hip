// Does ds_read ignore the high bits of its address VGPR?  (If it did, the binary32 value 2^23 + n could be used as the address n
// without a conversion.)  Prints what a read at n | 0x4B000000 returns next to a read at n.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        lds[i] = 0x1000u + (uint32_t)i;
    }
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)lds;  // LDS byte address of the array
    const uint32_t n = base + 4u * threadIdx.x;
    uint32_t plain, high23, high31, high17;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(plain) : "v"(n) : "memory");
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(high23) : "v"(n | 0x4B000000u) : "memory");
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(high31) : "v"(n | 0x80000000u) : "memory");
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(high17) : "v"(n | 0x00040000u) : "memory");
    out[4 * threadIdx.x + 0] = plain;
    out[4 * threadIdx.x + 1] = high23;
    out[4 * threadIdx.x + 2] = high31;
    out[4 * threadIdx.x + 3] = high17;
}
int main()
{
    uint32_t *d, h[4 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) {
        printf("lane %d: plain %08x  |0x4B000000 %08x  |0x80000000 %08x  |0x00040000 %08x\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
    }
    return 0;
}
