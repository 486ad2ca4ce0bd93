// micro-benchmark: VALU issue rates on gfx950 for f64/f32 ops (cycles per wave-instruction per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITERS 4096
template <int OP>
__global__ void kern(double* out, double a0, float f0) {
    double x0 = a0 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y0 = f0 + threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
    for (int i = 0; i < N_ITERS; ++i) {
#define R8(INS, T) \
        asm volatile(INS : "+v"(T##0) : "v"(T##1)); asm volatile(INS : "+v"(T##1) : "v"(T##2)); \
        asm volatile(INS : "+v"(T##2) : "v"(T##3)); asm volatile(INS : "+v"(T##3) : "v"(T##4)); \
        asm volatile(INS : "+v"(T##4) : "v"(T##5)); asm volatile(INS : "+v"(T##5) : "v"(T##6)); \
        asm volatile(INS : "+v"(T##6) : "v"(T##7)); asm volatile(INS : "+v"(T##7) : "v"(T##0));
        if (OP == 0) { R8("v_add_f64 %0, %0, %1", x) }
        if (OP == 1) { R8("v_max_f64 %0, %0, %1", x) }
        if (OP == 2) { R8("v_mul_f64 %0, %0, %1", x) }
        if (OP == 3) { R8("v_fma_f64 %0, %0, %1, %1", x) }
        if (OP == 4) { R8("v_add_f32 %0, %0, %1", y) }
        if (OP == 5) { R8("v_max_f32 %0, %0, %1", y) }
        if (OP == 6) { R8("v_fma_f32 %0, %0, %1, %1", y) }
        if (OP == 7) { R8("v_pk_add_f32 %0, %0, %1", x) }
        if (OP == 8) { R8("v_pk_mul_f32 %0, %0, %1", x) }
        if (OP == 9) { R8("v_pk_max_f32 %0, %0, %1", x) }   // may not exist
        if (OP == 10) { R8("v_min_f64 %0, %0, %1", x) }
        if (OP == 11) { R8("v_cmp_ge_f64 vcc, %0, %1", x) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}
template <int OP> void run(const char* name, double* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int wgs = 256 * 8;  // 8 WGs of 256 threads per CU -> 8 waves/SIMD
    kern<OP><<<wgs, 256>>>(d, 1.0, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kern<OP><<<wgs, 256>>>(d, 1.0, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts_per_simd = (double)wgs * 4 / 1024 * N_ITERS * 8;   // wave-instr per SIMD
    printf("%-14s %8.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
    run<0>("v_add_f64", d); run<1>("v_max_f64", d); run<10>("v_min_f64", d); run<2>("v_mul_f64", d); run<3>("v_fma_f64", d);
    run<11>("v_cmp_ge_f64", d);
    run<4>("v_add_f32", d); run<5>("v_max_f32", d); run<6>("v_fma_f32", d); run<7>("v_pk_add_f32", d); run<8>("v_pk_mul_f32", d);
#ifdef PKMAX
    run<9>("v_pk_max_f32", d);
#endif
    return 0;
}
