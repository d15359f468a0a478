#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* seed, double* one, double* two, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    seed[i] = y;
    double t = v * y, e = __builtin_fma(-t, y, 1.0), pp = __builtin_fma(0.375, e, 0.5), ye = y * e;
    y = __builtin_fma(ye, pp, y);
    one[i] = y;
    double t2 = v * y, e2 = __builtin_fma(-t2, y, 1.0);
    two[i] = __builtin_fma(y * e2, 0.5, y);
}
int main() {
    const int n = 1 << 20; double *hx = new double[n], *h0 = new double[n], *h1 = new double[n], *h2 = new double[n];
    for (int i = 0; i < n; ++i) hx[i] = exp((drand48() - 0.5) * 40.0);
    for (int i = 0; i < n / 2; ++i) hx[i] = 1.0 + drand48();
    double *dx, *d0, *d1, *d2; hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0, m2 = 0;
    for (int i = 0; i < n; ++i) { long double ex = 1.0L / sqrtl((long double)hx[i]);
        m0 = fmax(m0, fabs((double)((h0[i] - ex) / ex))); m1 = fmax(m1, fabs((double)((h1[i] - ex) / ex))); m2 = fmax(m2, fabs((double)((h2[i] - ex) / ex))); }
    printf("rsq seed max rel err %.3e  one cubic step %.3e  plus newton %.3e\n", m0, m1, m2);
}
