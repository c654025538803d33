// Developer check: is the unscaled square-root sequence bit-identical to the compiler's sqrt() on the device?
// The compiler's expansion of an f64 sqrt scales arguments below 2^-767 up and the result back down; for every
// other argument those steps multiply by 2^0.  Compares both over random bit patterns of the whole f64 range.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
__device__ __forceinline__ double sqrt_unscaled(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;  // +-0 and +inf come back as they are
}
__device__ __forceinline__ double sqrt_fast(double x) { return x < 0x1p-767 ? sqrt(x) : sqrt_unscaled(x); }
__global__ void k(const double* x, unsigned long long* bad, double* first, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = sqrt(x[i]), b = sqrt_fast(x[i]);
    if (__double_as_longlong(a) != __double_as_longlong(b) && !(a != a && b != b))
        if (atomicAdd(bad, 1ull) == 0) { first[0] = x[i]; first[1] = a; first[2] = b; }
}
int main() {
    const int n = 1 << 24;
    std::vector<double> x(n);
    std::mt19937_64 g(11);
    double *dx, *dfirst; unsigned long long *dbad;
    hipMalloc(&dx, n * 8); hipMalloc(&dbad, 8); hipMalloc(&dfirst, 24); hipMemset(dbad, 0, 8);
    for (int round = 0; round < 12; ++round) {
        for (int i = 0; i < n; ++i) {
            uint64_t bits = g();
            if (round >= 6) bits = (bits & 0x800fffffffffffffull) | ((uint64_t)(1023 - 40 + (int)(g() % 80)) << 52);  // near 1
            if (round % 6 == 5) bits &= 0x7fffffffffffffffull;                                                      // positive only
            std::memcpy(&x[i], &bits, 8);
        }
        x[0] = 0.0; x[1] = -0.0; x[2] = 1.0 / 0.0; x[3] = 0x1p-767; x[4] = 0x1.fffffffffffffp-768; x[5] = 4.9e-324; x[6] = 0x1p-1022;
        hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
        k<<<n / 256, 256>>>(dx, dbad, dfirst, n);
        hipDeviceSynchronize();
    }
    unsigned long long bad; double first[3];
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, dfirst, 24, hipMemcpyDeviceToHost);
    printf("sqrt_fast vs sqrt: %llu mismatches of %ld", bad, 12L * n);
    if (bad) printf(" (first: x=%a sqrt=%a fast=%a)", first[0], first[1], first[2]);
    printf("\n");
    return bad != 0;
}
