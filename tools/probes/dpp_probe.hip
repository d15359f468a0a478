#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    int lane = threadIdx.x;
    int v = lane;
    out[0 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    out[1 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    out[2 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x104, 0xf, 0xf, false);  // row_shl:4
    out[3 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);
    out[5 * 64 + lane] = __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);   // xor 4 attempt
    out[6 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[7 * 64 + lane] = r16[0]; out[8 * 64 + lane] = r16[1];
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    out[9 * 64 + lane] = r32[0]; out[10 * 64 + lane] = r32[1];
}
int main() {
    int* d; hipMalloc(&d, 11 * 64 * 4);
    probe<<<1, 64>>>(d);
    int h[11 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"qp1032", "qp2301", "row_shl4", "row_shr4", "row_ror8", "xor4", "half_mirror", "pl16[0]", "pl16[1]", "pl32[0]", "pl32[1]"};
    for (int r = 0; r < 11; ++r) { printf("%-12s", names[r]); for (int l = 0; l < 64; ++l) printf(" %d", h[r * 64 + l]); printf("\n"); }
    return 0;
}
