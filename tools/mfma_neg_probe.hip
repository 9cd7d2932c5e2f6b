// Does the blgp operand of v_mfma_f64_16x16x4 negate operands (neg:[a,b,c]) on gfx950?  Prints D[0][0] for a = 2, b = 3, c = 1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int BLGP>
__global__ void probe(double* out) {
    d4 c = {1.0, 1.0, 1.0, 1.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(2.0, 3.0, c, 0, 0, BLGP);
    if (threadIdx.x == 0) out[BLGP] = c[0];
}
int main() {
    double* d; hipMalloc(&d, 8 * sizeof(double)); hipMemset(d, 0, 64);
    probe<0><<<1, 64>>>(d); probe<1><<<1, 64>>>(d); probe<2><<<1, 64>>>(d); probe<3><<<1, 64>>>(d); probe<4><<<1, 64>>>(d);
    double h[8]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 5; ++i) printf("blgp %d -> %g   (plain: 2*3*4 + 1 = 25)\n", i, h[i]);
    return 0;
}
