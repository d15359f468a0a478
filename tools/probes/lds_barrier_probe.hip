// Cost of one "barrier + LDS round trip" step for a single workgroup (what bounds the inner Jacobi round and the
// in-LDS elimination): ITER steps of {barrier; 4 x ds_read_b64 (dependent on the previous write); fma; 4 x ds_write}.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_barrier_probe lds_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NB>
__global__ void __launch_bounds__(512) probe(int iters, double* out, long long* cyc) {
    __shared__ double s[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) s[i] = i * 1e-3;
    __syncthreads();
    const long long t0 = clock64();
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        const int o = (tid * 7 + it * 33) & 1023;
        const double a = s[o], b = s[o + 1024], c = s[o + 2048], d = s[o + 3072];
        const double y0 = 0.8 * a - 0.6 * b, y1 = 0.6 * a + 0.8 * b, y2 = 0.8 * c - 0.6 * d, y3 = 0.6 * c + 0.8 * d;
        if (NB >= 2) __syncthreads();
        s[o] = 0.8 * y0 - 0.6 * y2;
        s[o + 1024] = 0.6 * y0 + 0.8 * y2;
        s[o + 2048] = 0.8 * y1 - 0.6 * y3;
        s[o + 3072] = 0.6 * y1 + 0.8 * y3;
        acc += y0;
        if (NB >= 1) __syncthreads();
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * 512 * 64);
    hipMalloc(&cyc, sizeof(long long) * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int threads : {64, 256, 512})
        for (int nb = 0; nb < 3; ++nb)
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nb == 0) hipLaunchKernelGGL(probe<0>, dim3(8), dim3(threads), 0, 0, iters, out, cyc);
                if (nb == 1) hipLaunchKernelGGL(probe<1>, dim3(8), dim3(threads), 0, 0, iters, out, cyc);
                if (nb == 2) hipLaunchKernelGGL(probe<2>, dim3(8), dim3(threads), 0, 0, iters, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
                if (rep) printf("threads %3d barriers/step %d: %.3f us/step, %lld clock64 ticks/step\n", threads, nb, ms * 1e3 / iters, c / iters);
            }
    return 0;
}
