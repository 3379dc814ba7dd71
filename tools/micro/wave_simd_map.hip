// Which SIMD does each wave of a workgroup land on?  Prints HW_ID fields per wave for 256- and
// 512-thread workgroups (one workgroup per CU region sampled).
//   hipcc --offload-arch=gfx950 -O3 -o wave_simd_map.bin wave_simd_map.hip && ./wave_simd_map.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(unsigned* out) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
}

int main() {
    for (int threads : {256, 512}) {
        const int blocks = 512, wpb = threads / 64;
        unsigned* d;
        if (hipMalloc(&d, sizeof(unsigned) * blocks * wpb) != hipSuccess) return 1;
        k<<<blocks, threads>>>(d);
        std::vector<unsigned> h(blocks * wpb);
        if (hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        printf("workgroup of %d waves: wave -> (wave_id, simd, cu, sh, se) for the first 4 workgroups\n", wpb);
        for (int b = 0; b < 4; ++b) {
            for (int w = 0; w < wpb; ++w) {
                unsigned v = h[b * wpb + w];
                printf("  wg %d wave %d: slot %u simd %u cu %u sh %u se %u\n", b, w, v & 15, (v >> 4) & 3, (v >> 8) & 15,
                       (v >> 12) & 1, (v >> 13) & 7);
            }
        }
        // how often do waves w and w + wpb/2 share a SIMD, and waves 2i, 2i+1?
        int same_half = 0, same_adjacent = 0, n = 0;
        for (int b = 0; b < blocks; ++b)
            for (int w = 0; w < wpb / 2; ++w) {
                unsigned a = (h[b * wpb + w] >> 4) & 3, c = (h[b * wpb + w + wpb / 2] >> 4) & 3;
                unsigned e = (h[b * wpb + 2 * w] >> 4) & 3, f = (h[b * wpb + 2 * w + 1] >> 4) & 3;
                same_half += a == c; same_adjacent += e == f; ++n;
            }
        printf("  waves (w, w+%d) on one SIMD: %d of %d; waves (2i, 2i+1) on one SIMD: %d of %d\n", wpb / 2, same_half, n,
               same_adjacent, n);
        hipFree(d);
    }
    return 0;
}
