// which SIMD does wave w of a 512- or 1024-thread workgroup run on?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int wave = threadIdx.x >> 6;
    const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
    const int cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = simd | (cu << 8);
}
int main() {
    int* d; hipMalloc(&d, 64 * 16 * 4);
    for (int nt : {256, 512, 1024}) {
        hipMemset(d, 0xff, 64 * 16 * 4);
        hipLaunchKernelGGL(k, dim3(8), dim3(nt), 0, 0, d);
        int h[8 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 4; ++b) { printf("nt=%d block %d: ", nt, b); for (int w = 0; w < nt / 64; ++w) printf("%d ", h[b * 16 + w] & 255); printf(" (cu %d)\n", h[b*16] >> 8); }
    }
    return 0;
}
