// Probe (round 6): which bits of the third operand does v_alignbyte_b32 read on gfx950?  Prints the result for shift operands 0..40.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned s = threadIdx.x;
    o[s] = __builtin_amdgcn_alignbyte(0x88776655u, 0x44332211u, s);
}
int main() {
    unsigned *d, h[64];
    hipMalloc(&d, 256); k<<<1, 64>>>(d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 41; i++) printf("%2d %08x%s", i, h[i], i % 4 == 3 ? "\n" : "   ");
    printf("\n");
    int low2 = 1; for (int i = 0; i < 64; i++) if (h[i] != h[i & 3]) low2 = 0;
    printf("only bits [1:0] matter: %s\n", low2 ? "yes" : "no");
    return 0;
}
