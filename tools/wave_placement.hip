// Where the hardware puts the waves of a workgroup: SIMD and CU of every wave of 512-thread and 256-thread workgroups (HW_ID register).
// The producer / consumer kernels pair wave w with wave w + n/2 of a workgroup: they share a SIMD only if waves are dealt round the SIMDs.
// hipcc --offload-arch=gfx950 -O3 -o wave_placement tools/wave_placement.hip && ./wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *out, int spin) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned v = threadIdx.x;
    for (int i = 0; i < spin; i++) { v += 0x9e3779b9u; asm volatile("" : "+v"(v)); }  // (stay resident while the others arrive)
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
    if (v == 0x12345u && spin < 0) out[0] = v;
}
int main() {
    for (int threads : {512, 256, 1024}) {
        const int wpw = threads / 64, grid = 512;
        unsigned *d;
        hipMalloc(&d, sizeof(unsigned) * grid * wpw);
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, nullptr, d, 20000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(grid * wpw);
        hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
        printf("workgroups of %d threads: SIMD of waves 0..%d (first 6 workgroups), then how often wave w and wave w + n/2 share a SIMD\n", threads, wpw - 1);
        for (int b = 0; b < 6; b++) {
            printf("  wg %d: cu %2u se %u |", b, (h[b * wpw] >> 8) & 15, (h[b * wpw] >> 13) & 7);
            for (int w = 0; w < wpw; w++) printf(" %u", (h[b * wpw + w] >> 4) & 3);
            printf("\n");
        }
        int same = 0, total = 0, hist[4] = {0, 0, 0, 0};
        for (int b = 0; b < grid; b++) {
            for (int w = 0; w < wpw / 2; w++) {
                same += ((h[b * wpw + w] >> 4) & 3) == ((h[b * wpw + w + wpw / 2] >> 4) & 3);
                total++;
            }
            for (int w = 0; w < wpw; w++) hist[(h[b * wpw + w] >> 4) & 3]++;
        }
        printf("  pairs (w, w + n/2) on one SIMD: %d of %d; waves per SIMD id: %d %d %d %d\n", same, total, hist[0], hist[1], hist[2], hist[3]);
        hipFree(d);
    }
    return 0;
}
