// Probe: one wave that stays busy for a given time on a stream of its own, so that the device never idles between the short
// dependent launches of a broadband hop.  Question: do those launches run at a lower clock than a saturating kernel does?
#include <hip/hip_runtime.h>
__global__ void spin_kernel(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t;
    do {
        __builtin_amdgcn_s_sleep(16);
        t = __builtin_amdgcn_s_memtime();
    } while (t - t0 < ticks);                 // every wave reaches this exit: the time is bounded by the caller (<= 5 s)
    if (out && threadIdx.x == 0) *out = t - t0;
}
extern "C" int spin_start(double ms, int waves) {
    static hipStream_t s = nullptr;
    if (!s && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
    if (ms > 5000.0) ms = 5000.0;
    hipLaunchKernelGGL(spin_kernel, dim3(waves > 0 ? waves : 1), dim3(64), 0, s, (unsigned long long)(ms * 2.4e6), (unsigned long long*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int spin_wait() { return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }
