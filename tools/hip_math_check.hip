// Developer check: are the device's f64 sqrt and division bit-identical to the host's (correctly rounded) ones?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k_sqrt(const double* x, double* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = sqrt(x[i]); }
__global__ void k_div(const double* a, const double* b, double* y, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = a[i] / b[i]; }
int main() {
    const int n = 1 << 24;
    std::vector<double> x(n), b(n), y(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    double *dx, *db, *dy;
    hipMalloc(&dx, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dy, n * 8);
    long bad_sqrt = 0, bad_div = 0;
    for (int round = 0; round < 6; ++round) {
        double scale = round == 0 ? 1.0 : round == 1 ? 4.0 : round == 2 ? 1e-3 : round == 3 ? 300.0 : round == 4 ? 2.0 : 1e6;
        for (int i = 0; i < n; ++i) { x[i] = u(g) * scale + 1e-12; b[i] = (u(g) - 0.5) * 7.0 + 1e-9; }
        hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
        k_sqrt<<<n / 256, 256>>>(dx, dy, n); hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) if (y[i] != std::sqrt(x[i])) { if (bad_sqrt < 5) printf("sqrt(%a) gpu %a host %a\n", x[i], y[i], std::sqrt(x[i])); bad_sqrt++; }
        k_div<<<n / 256, 256>>>(dx, db, dy, n); hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) if (y[i] != x[i] / b[i]) { if (bad_div < 5) printf("div gpu %a host %a\n", y[i], x[i] / b[i]); bad_div++; }
    }
    printf("sqrt mismatches %ld, div mismatches %ld of %ld\n", bad_sqrt, bad_div, 6L * n);
    return 0;
}
