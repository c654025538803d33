// Measures what the FP64 vector ALUs of the GPU deliver with every wave slot busy: a chain-free stream of
// v_fma_f64 (2 flops per lane) and a stream of separate v_mul_f64 / v_add_f64 (1 flop each: what code built without
// multiply-add contraction can issue at best).  Built as raytracer_rs_amd/librt_peak.so; bench.py calls
// rt_fp64_peak() for the denominators of its roofline fractions.
#include <hip/hip_runtime.h>

#include <cstdio>

namespace {

constexpr int CHAINS = 16;   // independent accumulators per lane: far more than the FP64 pipeline is deep
constexpr int INNER = 64;    // unrolled operations per chain and loop round

template <bool FUSED>
__global__ __launch_bounds__(256) void peak_kernel(double *out, int rounds, double seed) {
    double a[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) a[k] = seed + (double)(threadIdx.x + k) * 1e-9;
    const double m = 1.0 + 1e-12 * seed, c = 1e-13 * seed;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int i = 0; i < INNER; ++i) {
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) {
                if (FUSED) {
                    a[k] = __builtin_fma(a[k], m, c);
                } else {
                    // one multiply and one add per pair of steps, never contracted (the file is built with
                    // -ffp-contract=off and the operations are on different chains' values)
                    a[k] = (i & 1) ? a[k] * m : a[k] + c;
                }
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) s += a[k];
    if (s == 123.456) out[threadIdx.x] = s;  // never true: keeps the arithmetic alive
}

template <bool FUSED>
int measure(double *tflops) {
    int device = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    double *out = nullptr;
    if (hipMalloc((void **)&out, 256 * sizeof(double)) != hipSuccess) return -1;
    const int blocks = prop.multiProcessorCount * 8;  // 8 x 4 waves per CU = 8 waves per SIMD
    const int rounds = 400;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    double best = 0.0;
    for (int rep = 0; rep < 6; ++rep) {  // the first repetitions bring the clocks up
        (void)hipEventRecord(e0, nullptr);
        peak_kernel<FUSED><<<blocks, 256, 0, nullptr>>>(out, rounds, 1.0 + rep);
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) return -1;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 256.0 * rounds * INNER * CHAINS * (FUSED ? 2.0 : 1.0);
        const double tf = flops / (ms * 1e-3) / 1e12;
        if (tf > best) best = tf;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    *tflops = best;
    return 0;
}

}  // namespace

// TFLOP/s of a saturating v_fma_f64 stream (fma_tflops, an FMA counted as 2) and of a v_mul_f64 / v_add_f64 stream
// (unfused_tflops) on the current device.  Returns 0, or -1 when the device cannot be used.
extern "C" int rt_fp64_peak(double *fma_tflops, double *unfused_tflops) {
    if (!fma_tflops || !unfused_tflops) return -1;
    if (measure<true>(fma_tflops) != 0) return -1;
    return measure<false>(unfused_tflops);
}

#ifdef RT_PEAK_MAIN
int main() {
    double f = 0, u = 0;
    if (rt_fp64_peak(&f, &u) != 0) return 1;
    std::printf("fp64 vector peak measured: fma %.2f TFLOP/s, mul/add unfused %.2f TFLOP/s\n", f, u);
    return 0;
}
#endif
