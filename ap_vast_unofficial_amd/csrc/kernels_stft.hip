// K2-K4: batched sine-window analysis STFT and synthesis + overlap-add.
//
//   analysis : spec[c][k] = rfft(window * x[c][:])[k]                 reference Python/apvast.py:202-203,
//                                                                     246-255, 430-431
//   synthesis: new = window * irfft(spec[c]); overlap[c] = shift(overlap[c], H) + new;
//              out[c][0:H] = overlap[c][0:H]                          apvast.py:212-225, 265-293, 457-504
//
// One workgroup per channel.  A length-N real transform is done as an N/2-point complex
// radix-2 FFT in LDS (N <= 8192 -> <= 32 KiB) plus the even/odd split step.  Twiddles and
// the window come from tables computed once on the host in double precision.
#include "apv_internal.h"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int STFT_TPB = 256;
constexpr int STFT_MAX_N = 8192;

struct Tables {
    float2* tw = nullptr;     // exp(-2 pi i j / N), j < N/2
    float* win = nullptr;     // sin(pi n / N), n < N     (apvast.py:94)
};

std::mutex g_tab_mu;
std::map<std::pair<int, int>, Tables> g_tabs;   // (device, N) -> tables

hipError_t get_tables(int N, Tables* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto it = g_tabs.find({dev, N});
    if (it != g_tabs.end()) {
        *out = it->second;
        return hipSuccess;
    }
    std::vector<float2> tw(N / 2);
    std::vector<float> win(N);
    const double PI = 3.14159265358979323846;
    for (int j = 0; j < N / 2; ++j) {
        tw[j].x = (float)std::cos(-2.0 * PI * j / N);
        tw[j].y = (float)std::sin(-2.0 * PI * j / N);
    }
    for (int n = 0; n < N; ++n) win[n] = (float)std::sin(PI * n / N);
    Tables t;
    e = hipMalloc(&t.tw, sizeof(float2) * (N / 2));
    if (e != hipSuccess) return e;
    e = hipMalloc(&t.win, sizeof(float) * N);
    if (e != hipSuccess) return e;
    e = hipMemcpy(t.tw, tw.data(), sizeof(float2) * (N / 2), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    e = hipMemcpy(t.win, win.data(), sizeof(float) * N, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    g_tabs[{dev, N}] = t;
    *out = t;
    return hipSuccess;
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-place radix-2 DIT on bit-reversed data in LDS.  Nh = 2^lg points; tw indexed with stride N/Nh = 2.
__device__ __forceinline__ void fft_lds(float2* z, int Nh, int lg, const float2* __restrict__ tw, int N) {
    const int tid = threadIdx.x;
    for (int s = 1; s <= lg; ++s) {
        const int half = 1 << (s - 1);
        const int tstride = N >> s;                       // N / (2*half)
        for (int b = tid; b < Nh / 2; b += STFT_TPB) {
            const int grp = b >> (s - 1), pos = b & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            const float2 w = tw[pos * tstride];
            const float2 u = z[i0], v = cmulf(z[i1], w);
            z[i0] = make_float2(u.x + v.x, u.y + v.y);
            z[i1] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int bitrev(int v, int lg) { return (int)(__brev((unsigned)v) >> (32 - lg)); }

// spec element (c, k) is stored at spec[c * stride_c + k * stride_k]
// x is a ring: logical sample n of channel c lives at x[c*N + ((n + ring_off) & (N-1))]
__global__ void __launch_bounds__(STFT_TPB) stft_analysis_kernel(int N, int lg, const float* __restrict__ x,
                                                                 int ring_off, float2* __restrict__ spec,
                                                                 long stride_c, long stride_k,
                                                                 const float2* __restrict__ tw,
                                                                 const float* __restrict__ win) {
    extern __shared__ float2 z[];
    const int Nh = N >> 1;
    const int c = blockIdx.x, tid = threadIdx.x;
    const float* xin = x + (size_t)c * N;
    const float2* w2 = reinterpret_cast<const float2*>(win);
    const int mask = N - 1;
    for (int n = tid; n < Nh; n += STFT_TPB) {
        const float2 w = w2[n];
        const float v0 = xin[(2 * n + ring_off) & mask], v1 = xin[(2 * n + 1 + ring_off) & mask];
        z[bitrev(n, lg)] = make_float2(v0 * w.x, v1 * w.y);
    }
    __syncthreads();
    fft_lds(z, Nh, lg, tw, N);
    // even/odd split: X[k] = E[k] + e^{-2 pi i k/N} O[k]
    float2* out = spec + (size_t)c * stride_c;
    for (int k = tid; k <= Nh; k += STFT_TPB) {
        const float2 a = z[k == Nh ? 0 : k];
        const float2 bq = z[k == 0 ? 0 : Nh - k];
        const float2 b = make_float2(bq.x, -bq.y);                       // conj(Z[Nh-k])
        const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 dm = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 o = make_float2(dm.y, -dm.x);                       // -i * dm
        const float2 wk = (k == Nh) ? make_float2(-1.f, 0.f) : tw[k];
        const float2 ow = cmulf(o, wk);
        out[(size_t)k * stride_k] = make_float2(e.x + ow.x, e.y + ow.y);
    }
}

__global__ void __launch_bounds__(STFT_TPB) istft_ola_kernel(int N, int lg, int H, const float2* __restrict__ spec,
                                                             long stride_c, long stride_k,
                                                             float* __restrict__ overlap, float* __restrict__ out,
                                                             const float2* __restrict__ tw,
                                                             const float* __restrict__ win) {
    extern __shared__ float2 z[];
    const int Nh = N >> 1;
    const int c = blockIdx.x, tid = threadIdx.x;
    const float2* X = spec + (size_t)c * stride_c;
    // Z[k] = E[k] + i O[k];  inverse transform as conj(FFT(conj(Z))) / Nh
    for (int k = tid; k < Nh; k += STFT_TPB) {
        float2 a = X[(size_t)k * stride_k];
        float2 bq = X[(size_t)(Nh - k) * stride_k];
        if (k == 0) {                                                   // irfft drops imag of DC and Nyquist
            a.y = 0.f;
            bq.y = 0.f;
        }
        const float2 b = make_float2(bq.x, -bq.y);
        const float2 e = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 dm = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 wk = tw[k];
        const float2 o = cmulf(dm, make_float2(wk.x, -wk.y));            // * e^{+2 pi i k/N}
        const float2 Z = make_float2(e.x - o.y, e.y + o.x);              // E + i O
        z[bitrev(k, lg)] = make_float2(Z.x, -Z.y);                       // conj
    }
    __syncthreads();
    fft_lds(z, Nh, lg, tw, N);
    const float scale = 1.0f / (float)Nh;
    float* ov = overlap + (size_t)c * N;
    float* zf = reinterpret_cast<float*>(z);
    const float2* w2 = reinterpret_cast<const float2*>(win);
    // windowed new block, in place: x[2n] = Re z[n], x[2n+1] = -Im(conj-FFT)[n]
    for (int n = tid; n < Nh; n += STFT_TPB) {
        const float2 v = z[n], w = w2[n];
        z[n] = make_float2(v.x * scale * w.x, -v.y * scale * w.y);
    }
    __syncthreads();
    for (int n = tid; n < N; n += STFT_TPB) {
        const float old = (n < N - H) ? ov[n + H] : 0.f;
        zf[n] += old;
    }
    __syncthreads();
    for (int n = tid; n < N; n += STFT_TPB) ov[n] = zf[n];
    if (out != nullptr)
        for (int n = tid; n < H; n += STFT_TPB) out[(size_t)c * H + n] = zf[n];
}

int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace

static bool stft_size_ok(int N, std::string* why) {
    if (N < 8 || N > STFT_MAX_N || (N & (N - 1)) != 0) {
        if (why) *why = "STFT block size must be a power of two in [8, 8192]";
        return false;
    }
    return true;
}

hipError_t apv_launch_stft_analysis_strided(int N, int n_ch, const float* x, int ring_off, float2* spec,
                                            long stride_c, long stride_k, hipStream_t s, std::string* why) {
    if (!stft_size_ok(N, why)) return hipErrorInvalidValue;
    if (n_ch <= 0) return hipSuccess;
    Tables t;
    hipError_t e = get_tables(N, &t);
    if (e != hipSuccess) return e;
    const int Nh = N / 2;
    hipLaunchKernelGGL(stft_analysis_kernel, dim3(n_ch), dim3(STFT_TPB), sizeof(float2) * Nh, s, N, ilog2(Nh), x,
                       ring_off & (N - 1), spec, stride_c, stride_k, t.tw, t.win);
    return hipGetLastError();
}

hipError_t apv_launch_istft_ola_strided(int N, int H, int n_ch, const float2* spec, long stride_c, long stride_k,
                                        float* overlap, float* out, hipStream_t s, std::string* why) {
    if (!stft_size_ok(N, why)) return hipErrorInvalidValue;
    if (H <= 0 || H > N) {
        if (why) *why = "hop size out of range";
        return hipErrorInvalidValue;
    }
    if (n_ch <= 0) return hipSuccess;
    Tables t;
    hipError_t e = get_tables(N, &t);
    if (e != hipSuccess) return e;
    const int Nh = N / 2;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(n_ch), dim3(STFT_TPB), sizeof(float2) * Nh, s, N, ilog2(Nh), H, spec,
                       stride_c, stride_k, overlap, out, t.tw, t.win);
    return hipGetLastError();
}

hipError_t apv_launch_stft_analysis(int N, int n_ch, const float* x, float2* spec, hipStream_t s, std::string* why) {
    return apv_launch_stft_analysis_strided(N, n_ch, x, 0, spec, N / 2 + 1, 1, s, why);
}

hipError_t apv_launch_istft_ola(int N, int H, int n_ch, const float2* spec, float* overlap, float* out,
                                hipStream_t s, std::string* why) {
    return apv_launch_istft_ola_strided(N, H, n_ch, spec, N / 2 + 1, 1, overlap, out, s, why);
}
