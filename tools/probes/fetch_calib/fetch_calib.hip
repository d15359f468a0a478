// What does rocprofv3's FETCH_SIZE report for the access pattern of the order-16 update kernel?  MI355X_MICROARCH.md calibrates
// the counter for 16-byte-per-lane streaming reads (it reports exactly half the bytes) and calls other widths uncalibrated.  The
// kernel's slab reads are 8 bytes per lane: lane l of a wave reads element 64 s + l of its own 4 KB slab, s = 0..7 (correlate16).
// This program reads a buffer of KNOWN size once in exactly that pattern (`slab8`), once in the guide's pattern (`stream16`: 16
// bytes per lane, consecutive waves consecutive kilobytes) and once as 8-byte-per-lane streaming (`stream8`); run it under
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_calib
// and divide.  Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(64) slab8(const float2* __restrict__ x, float* __restrict__ out, int slab_elems) {
    const float2* s = x + (size_t)blockIdx.x * slab_elems;
    float acc = 0.f;
    for (int e = threadIdx.x; e < slab_elems; e += 64) {
        const float2 v = s[e];
        acc += v.x + v.y;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;            // never true: keeps the loads
}
__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ x, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = x[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) stream8(const float2* __restrict__ x, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float2 v = x[i];
        acc += v.x + v.y;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;                   // 1 GiB: four times the Infinity Cache
    void* buf;
    float* out;
    hipMalloc(&buf, bytes);
    hipMalloc((void**)&out, 4 << 20);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const int slab_elems = 512;                             // 4 KB of float2: one 32 x 16 control-point matrix
    const int n_slabs = (int)(bytes / (slab_elems * sizeof(float2)));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(slab8, dim3(n_slabs), dim3(64), 0, 0, (const float2*)buf, out, slab_elems);
        hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, (const float4*)buf, out, bytes / 16);
        hipLaunchKernelGGL(stream8, dim3(8192), dim3(256), 0, 0, (const float2*)buf, out, bytes / 8);
    }
    hipDeviceSynchronize();
    printf("bytes_per_kernel %zu\n", bytes);
    return 0;
}
