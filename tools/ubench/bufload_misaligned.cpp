// Does a 16-byte raw buffer load at a 4-byte-aligned (not 16-byte-aligned) offset return the four dwords at
// that offset?  hipcc --offload-arch=gfx950 bufload_misaligned.cpp -o bufload_misaligned && ./bufload_misaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, unsigned bytes, float* out, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, bytes, 0x00020000);
    const int lane = threadIdx.x;
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(16 * lane + shift), 0, 0);
    out[4 * lane + 0] = __builtin_bit_cast(float, v.x);
    out[4 * lane + 1] = __builtin_bit_cast(float, v.y);
    out[4 * lane + 2] = __builtin_bit_cast(float, v.z);
    out[4 * lane + 3] = __builtin_bit_cast(float, v.w);
}
int main() {
    const int N = 1024;
    float h[N], *d, *o, ho[256];
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
    for (int shift : {0, 4, 8, 12}) {
        k<<<1, 64>>>(d, N * 4, o, shift);
        hipMemcpy(ho, o, 256 * 4, hipMemcpyDeviceToHost);
        printf("shift %2d: lane0 %g %g %g %g | lane1 %g %g %g %g | lane5 %g %g %g %g\n", shift, ho[0], ho[1], ho[2], ho[3], ho[4], ho[5],
               ho[6], ho[7], ho[20], ho[21], ho[22], ho[23]);
    }
    return 0;
}
