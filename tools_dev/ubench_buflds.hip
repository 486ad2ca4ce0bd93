// checks that an out-of-range lane of buffer_load_dwordx4 ... lds writes ZEROS to its LDS slot (not "no write")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned short* x, unsigned nbytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16u;
    if (threadIdx.x & 1) off = 0xfffffff0u;
    if (threadIdx.x == 2) off = nbytes - 8;          // straddles the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)smem, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = reinterpret_cast<unsigned*>(smem)[i];
}
int main() {
    std::vector<unsigned short> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)(i + 1);
    unsigned short* d; unsigned* o;
    hipMalloc(&d, 8192); hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 8192u, o);
    std::vector<unsigned> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        unsigned v = r[l * 4 + j];
        unsigned e0 = (l * 8 + j * 2 + 1), e = (e0 | ((e0 + 1) << 16));
        if (l & 1) e = 0;
        if (l == 2) { printf("lane2 word%d = %08x\n", j, v); continue; }
        if (v != e) { if (bad < 8) printf("lane %d word %d: %08x expected %08x\n", l, j, v, e); ++bad; }
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
