// Probes the operand layout and index encoding of v_smfmac_f32_16x16x64_bf16 on gfx950 (no public doc in the image).
// Probe 1: B register j of lane (gq, li) holds the label 16 gq + j + 1; A (compressed) is one-hot: slot i of the lanes with
//   lane >> 4 == G0 is 1.0, index pattern P in every group -> every C element = label of the B register that A slot multiplies.
// Probe 2: A one-hot in ONE lane (row): which C lanes / elements light up.
// build: hipcc -O3 --offload-arch=gfx950 tools_dev/smfmac_layout_check.hip -o tools_dev/ubench_smfmac_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(__bf16)))) __bf16 bf16x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

__global__ void probe1(int G0, int slot, int idx, float* C) {
    const int lane = threadIdx.x, gq = lane >> 4;
    bf16x8_t a;
    bf16x16_t b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)((gq == G0 && i == slot) ? 1.f : 0.f);
    for (int j = 0; j < 16; ++j) b[j] = (__bf16)(float)(16 * gq + j + 1);
    f32x4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, idx, 0, 0);
    for (int j = 0; j < 4; ++j) C[lane * 4 + j] = c[j];
}
__global__ void probe2(int L0, float* C) {
    const int lane = threadIdx.x;
    bf16x8_t a;
    bf16x16_t b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)((lane == L0 && i == 0) ? 1.f : 0.f);
    for (int j = 0; j < 16; ++j) b[j] = (__bf16)(float)((lane & 15) + 1);          // label = column
    f32x4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_smfmac_f32_16x16x64_bf16(a, b, c, 0x4444, 0, 0);
    for (int j = 0; j < 4; ++j) C[lane * 4 + j] = c[j];
}
int main() {
    float* dC; hipMalloc(&dC, 256 * 4);
    float h[256];
    const int pats[] = {0x4444, 0xEEEE, 0x8888, 0xCCCC, 0x9999, 0xDDDD, 0x44440000, 0x4E4E};
    for (int p = 0; p < 8; ++p) {
        printf("index pattern 0x%08x: label of the B register (16 gq + j + 1) hit by A slot i of lane group G0\n", pats[p]);
        for (int G0 = 0; G0 < 4; ++G0) {
            printf("  G0=%d:", G0);
            for (int s = 0; s < 8; ++s) {
                probe1<<<1, 64>>>(G0, s, pats[p], dC);
                hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost);
                bool same = true;
                for (int i = 1; i < 256; ++i) same = same && h[i] == h[0];
                printf(" %5.0f%s", h[0], same ? "" : "*");
            }
            printf("\n");
        }
    }
    for (int L0 : {0, 5, 16, 37}) {
        probe2<<<1, 64>>>(L0, dC);
        hipMemcpy(h, dC, sizeof(h), hipMemcpyDeviceToHost);
        printf("A one-hot in lane %d (slot 0): nonzero C at", L0);
        for (int i = 0; i < 256; ++i) if (h[i] != 0.f) printf(" [lane %d el %d]=%g", i / 4, i % 4, h[i]);
        printf("\n");
    }
    return 0;
}
