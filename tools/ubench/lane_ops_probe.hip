#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r32[0]; out[64 + lane] = r32[1];
    auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + lane] = r16[0]; out[192 + lane] = r16[1];
    // ds_bpermute: dst[i] = src[addr[i]/4]
    out[256 + lane] = __builtin_amdgcn_ds_bpermute((int)(((lane & 48) | 3) * 4), (int)a);
    // row_ror:15 via update_dpp (0x12F)
    out[320 + lane] = __builtin_amdgcn_update_dpp(0, (int)a, 0x12F, 0xF, 0xF, false);
    // row_shr:1 without bound_ctrl, old = 999
    out[384 + lane] = __builtin_amdgcn_update_dpp(999, (int)a, 0x111, 0xF, 0xF, false);
    out[448 + lane] = __builtin_amdgcn_update_dpp(999, (int)a, 0x101, 0xF, 0xF, false);
}
int main() {
    unsigned* d; hipMalloc(&d, 512 * 4);
    k<<<1, 64>>>(d);
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"swap32.dst(a)", "swap32.src(b)", "swap16.dst(a)", "swap16.src(b)", "bpermute row|3", "row_ror:15", "row_shr:1 old999", "row_shl:1 old999"};
    for (int t = 0; t < 8; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; ++i) printf(" %u", h[t * 64 + i]); printf("\n"); }
    return 0;
}
