// Probe of ds_read_b64_tr_b16 (gfx950): which element does lane i receive from which lane's address?
// Build: hipcc --offload-arch=gfx950 -O2 tools/microbench/tr_read_probe.hip -o tools/microbench/tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 v4h __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int row_stride_elems, int row_step) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;       // element value = its index
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // block of group grp: rows r0 + q * row_step (q = 0..3), columns c0 + 4p .. 4p + 3 of an image with row_stride_elems per row
    const int r0 = 8 * grp, c0 = 0;
    const unsigned short* a = lds + (r0 + q * row_step) * row_stride_elems + c0 + 4 * p;
    v4h r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) v4h*)a);
    const unsigned short* rs = (const unsigned short*)&r;
    for (int e = 0; e < 4; e++) out[lane * 4 + e] = rs[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int step = 1; step <= 2; step++) {
        probe<<<1, 64>>>(d, 72, step);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; lane++) {
            const int grp = lane >> 4, i = lane & 15;
            for (int e = 0; e < 4; e++) {
                const int want = (8 * grp + e * step) * 72 + i;       // row q = e of the block, column i
                if (h[lane * 4 + e] != want) { if (bad < 8) printf("step %d lane %d elem %d: got %d want %d\n", step, lane, e, h[lane * 4 + e], want); bad++; }
            }
        }
        printf("row step %d: %s (%d mismatches)\n", step, bad ? "DIFFERENT from the guide's description" : "as the guide describes", bad);
    }
    return 0;
}
