// Micro-benchmark (gfx950): issue rate / dependent latency of v_mfma_f64_16x16x4 and of v_fma_f64, alone and
// side by side on one SIMD, v_readlane and v_rsq_f64 chains.  Build: hipcc --offload-arch=gfx950 -O3 -o
// /tmp/mfma_f64_bench tools/mfma_f64_bench.hip ; numbers quoted in DESIGN.md §4.3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));

#define N_IT 2000

// mode 0: one MFMA dependent chain; 1: four independent MFMA chains; 2: FMA f64 8 independent chains (per lane)
// 3: waves with (wid & 1) run MFMA (4 chains), the others FMA (8 chains) -- two waves per SIMD
// 4: dependent v_fma_f64 chain; 5: readlane -> fma chain; 6: rsq chain
__global__ void __launch_bounds__(512) bench(int mode, double* out, long long* cyc) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double f[8];
    for (int i = 0; i < 8; ++i) f[i] = 1e-3 * (lane + i);
    __syncthreads();
    const long long t0 = clock64();
    int m = mode;
    if (mode == 3) m = (wid & 4) ? 1 : 2;  // waves 0-3 -> SIMD 0-3 FMA, waves 4-7 -> MFMA (each SIMD gets one of each)
    if (m == 0) {
        for (int i = 0; i < N_IT; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    } else if (m == 1) {
        for (int i = 0; i < N_IT / 4; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    } else if (m == 2) {
        for (int i = 0; i < N_IT / 8; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = fma(f[k], a, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = fma(f[k], b, a);
        }
    } else if (m == 4) {
        for (int i = 0; i < N_IT; ++i) f[0] = fma(f[0], a, b);
    } else if (m == 5) {
        for (int i = 0; i < N_IT; ++i) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(f[0]), 7);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(f[0]), 7);
            f[0] = fma(f[1], __hiloint2double(hi, lo), b);
        }
    } else if (m == 6) {
        for (int i = 0; i < N_IT; ++i) f[0] = __builtin_amdgcn_rsq(f[0]) + a;
    }
    const long long t1 = clock64();
    double s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

int main() {
    double* out;
    long long* cyc;
    hipMalloc(&out, sizeof(double) * 512 * 1024);
    hipMalloc(&cyc, sizeof(long long) * 8 * 1024);
    long long h[8];
    const char* names[] = {"mfma dependent chain", "mfma 4 independent chains", "fma f64 8 chains", "mixed: waves 0-3 fma, 4-7 mfma",
                           "fma f64 dependent chain", "readlane x2 -> fma chain", "rsq f64 + add chain"};
    for (int waves = 1; waves <= 8; waves *= 2) {
        for (int mode = 0; mode < 7; ++mode) {
            if (mode == 3 && waves != 8) continue;
            hipLaunchKernelGGL(bench, dim3(1), dim3(64 * waves), 0, 0, mode, out, cyc);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("waves/block %d  %-32s:", waves, names[mode]);
            for (int w = 0; w < waves; ++w) printf(" %.1f", (double)h[w] / N_IT);
            printf("  (clock64 ticks per op; mixed: per op of own kind)\n");
        }
    }
    // wall-clock: full chip, 8 waves per CU, MFMA 4 chains
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 1; mode <= 2; ++mode) {
        hipLaunchKernelGGL(bench, dim3(1024), dim3(512), 0, 0, mode, out, cyc);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(bench, dim3(1024), dim3(512), 0, 0, mode, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = 10.0 * 1024 * 8 * N_IT;
        const double flop = mode == 1 ? ops * 2048 : ops * 128;
        printf("%s full chip: %.3f ms, %.1f TFLOP/s\n", names[mode], ms, flop / ms * 1e-9);
    }
    return 0;
}
