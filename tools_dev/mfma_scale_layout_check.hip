// Probes the operand layout and the scale semantics of v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (OCP e4m3) operands on gfx950.
// Probe 1: A one-hot (byte slot i of the lanes with lane >> 4 == G0 is 1.0), B byte j of lane group gq carries a small label; three
//          labelings (gq + 1, (j & 7) + 1, (j >> 3) + 1) identify the B byte each A slot multiplies.
// Probe 2: A = B = ones, scale_a = 2.0 in ONE lane, 1.0 elsewhere: which C elements change, and by how much (32 of 128 products?).
// Probe 3: the same for scale_b.  Probe 4: opsel selects the scale byte.
// build: hipcc -O3 --offload-arch=gfx950 tools_dev/mfma_scale_layout_check.hip -o tools_dev/ubench_mfma_scale_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __host__ inline unsigned char e4m3_small(int v) {         // 0..8 exactly
    const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return t[v];
}
__global__ void probe1(int G0, int slot, int labeling, float* C) {
    const int lane = threadIdx.x, gq = lane >> 4;
    unsigned char a[32], b[32];
    for (int i = 0; i < 32; ++i) {
        a[i] = (gq == G0 && i == slot) ? 0x38 : 0x00;
        const int lab = labeling == 0 ? gq + 1 : (labeling == 1 ? (i & 7) + 1 : (i >> 3) + 1);
        b[i] = e4m3_small(lab);
    }
    v8i av, bv;
    memcpy(&av, a, 32); memcpy(&bv, b, 32);
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int j = 0; j < 4; ++j) C[lane * 4 + j] = c[j];
}
__global__ void probe2(int L0, int which, int opsel, unsigned sc, float* C) {
    const int lane = threadIdx.x;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = 0x38383838; bv[i] = 0x38383838; }
    const int sa = (which == 0 && lane == L0) ? (int)sc : 0x7f7f7f7f;
    const int sb = (which == 1 && lane == L0) ? (int)sc : 0x7f7f7f7f;
    v4f c = {0, 0, 0, 0};
    if (opsel == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa, 0, sb);
    else if (opsel == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 1, sa, 1, sb);
    else if (opsel == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 2, sa, 2, sb);
    else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 3, sa, 3, sb);
    for (int j = 0; j < 4; ++j) C[lane * 4 + j] = c[j];
}
int main() {
    float* dC; hipMalloc(&dC, 1024);
    float h[3][256];
    printf("A slot (G0, i) multiplies B byte (gq, j):\n");
    for (int G0 = 0; G0 < 4; ++G0) {
        printf(" G0=%d:", G0);
        for (int s = 0; s < 32; ++s) {
            for (int lab = 0; lab < 3; ++lab) { probe1<<<1, 64>>>(G0, s, lab, dC); hipMemcpy(h[lab], dC, 1024, hipMemcpyDeviceToHost); }
            bool same = true;
            for (int i = 1; i < 256; ++i) same = same && h[0][i] == h[0][0] && h[1][i] == h[1][0] && h[2][i] == h[2][0];
            printf(" (%d,%d)%s", (int)h[0][0] - 1, ((int)h[2][0] - 1) * 8 + (int)h[1][0] - 1, same ? "" : "*");
        }
        printf("\n");
    }
    for (int which = 0; which < 2; ++which)
        for (int L0 : {0, 21, 37, 63}) {
            probe2<<<1, 64>>>(L0, which, 0, 0x7f7f7f80u, dC);      // byte 0 = 128 -> x2
            hipMemcpy(h[0], dC, 1024, hipMemcpyDeviceToHost);
            printf("scale_%c = 2.0 (byte 0) in lane %d: changed C elements:", which ? 'b' : 'a', L0);
            int n = 0;
            for (int i = 0; i < 256; ++i) if (h[0][i] != 128.f) { if (n < 6) printf(" [lane %d el %d]=%g", i / 4, i % 4, h[0][i]); ++n; }
            printf("  (%d changed)\n", n);
        }
    for (int op = 0; op < 4; ++op) {
        probe2<<<1, 64>>>(5, 0, op, 0x83828180u, dC);              // bytes 0..3 = 128, 129, 130, 131 -> x2, x4, x8, x16
        hipMemcpy(h[0], dC, 1024, hipMemcpyDeviceToHost);
        float mx = 0; for (int i = 0; i < 256; ++i) mx = h[0][i] > mx ? h[0][i] : mx;
        printf("opsel %d, scale_a bytes {x2,x4,x8,x16} in lane 5: max C = %g (128 + 32 (s - 1))\n", op, mx);
    }
    return 0;
}
