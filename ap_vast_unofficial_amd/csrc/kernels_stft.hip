// K2-K4: batched sine-window analysis STFT and synthesis + overlap-add, float or double.
//
//   analysis : spec[c][k] = rfft(window * x[c][:])[k]                 reference Python/apvast.py:202-203,
//                                                                     246-255, 430-431; un-windowed,
//                                                                     zero-padded form for apvast.py:417-422
//   synthesis: new = window * irfft(spec[c]); overlap[c] = shift(overlap[c], H) + new;
//              out[c][0:H] = overlap[c][0:H]                          apvast.py:212-225, 265-293, 457-504
//
// One workgroup per channel.  A length-N real transform (N even) is an N/2-point complex Stockham FFT in LDS
// (mixed radix 4/2/3/5/7, so the reference's own N = 1600 works) plus the even/odd split step.  Twiddles and
// the window come from tables computed once per (device, N, dtype) on the host in double precision.
#include "apv_internal.h"
#include <cstdio>
#include <cstdlib>

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

constexpr int STFT_TPB = 256;
constexpr int STFT_MAX_N = 8192;
constexpr int MAX_STAGES = 14;

struct FftPlan {
    int N;                 // real length
    int Nh;                // complex length N/2
    int nstages;
    int radix[MAX_STAGES];
    int inplace;           // every stage is radix 4 or 2 and short enough for stockham_stage_inplace: ONE LDS buffer of Nh
    int max_it;            // butterflies per thread of the widest in-place stage (kernels are instantiated for 1 and for INPLACE_MAX_IT)
    int debug;             // timing aids of the analysis transforms (APV_STFT_DEBUG; results are wrong): 1 no spectrum stores, 2 no sample loads, 4 no stages
};
constexpr int INPLACE_MAX_IT = 4;      // butterflies per thread and stage the in-place form holds in registers

// 16-byte alignment for the double-precision pair: the compiler then moves it as ONE ds_read_b128 / ds_write_b128 / dwordx4
// (4 LDS cycles per wave-instruction where the two ds_read2_b64 halves of an 8-byte-aligned pair cost 16, and sixteen lanes a
// group at a 16-byte stride are a 2-way bank conflict on top: SQ_LDS_BANK_CONFLICT was 61 % of the LDS-active cycles of every
// transform kernel, profiles/r03/analysis_counters.md)
template <typename T> struct alignas(2 * sizeof(T)) C2 { T x, y; };
template <typename T> __device__ __forceinline__ C2<T> c2(T a, T b) { C2<T> r; r.x = a; r.y = b; return r; }
template <typename T> __device__ __forceinline__ C2<T> cadd(C2<T> a, C2<T> b) { return c2<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ C2<T> csub(C2<T> a, C2<T> b) { return c2<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ C2<T> cmul(C2<T> a, C2<T> b) {
    return c2<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <typename T>
struct Tables {
    C2<T>* tw = nullptr;   // exp(-2 pi i j / N), j < N (full circle)
    T* win = nullptr;      // sin(pi n / N), n < N     (apvast.py:94)
};

std::mutex g_tab_mu;
std::map<std::tuple<int, int, int>, std::pair<void*, void*>> g_tabs;   // (device, N, is_f64) -> (tw, win)

template <typename T>
hipError_t get_tables(int N, Tables<T>* out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    const auto key = std::make_tuple(dev, N, (int)(sizeof(T) == 8));
    auto it = g_tabs.find(key);
    if (it != g_tabs.end()) {
        out->tw = (C2<T>*)it->second.first;
        out->win = (T*)it->second.second;
        return hipSuccess;
    }
    std::vector<C2<T>> tw(N);
    std::vector<T> win(N);
    const double PI = 3.14159265358979323846;
    for (int j = 0; j < N; ++j) {
        tw[j].x = (T)std::cos(-2.0 * PI * j / N);
        tw[j].y = (T)std::sin(-2.0 * PI * j / N);
        win[j] = (T)std::sin(PI * j / N);
    }
    void *dtw = nullptr, *dwin = nullptr;
    e = hipMalloc(&dtw, sizeof(C2<T>) * N);
    if (e != hipSuccess) return e;
    e = hipMalloc(&dwin, sizeof(T) * N);
    if (e != hipSuccess) return e;
    // The tables are shared by every handle of the device, so they go up on a stream of their own that is drained before the
    // table is published: no stream of any handle can see them half written, and nothing rides on the null stream.
    hipStream_t up = nullptr;
    e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(dtw, tw.data(), sizeof(C2<T>) * N, hipMemcpyHostToDevice, up);
    if (e == hipSuccess) e = hipMemcpyAsync(dwin, win.data(), sizeof(T) * N, hipMemcpyHostToDevice, up);
    if (e == hipSuccess) e = hipStreamSynchronize(up);
    (void)hipStreamDestroy(up);
    if (e != hipSuccess) return e;
    g_tabs[key] = {dtw, dwin};
    out->tw = (C2<T>*)dtw;
    out->win = (T*)dwin;
    return hipSuccess;
}

// ---- Stockham stages ---------------------------------------------------------------------------
// One stage of radix R: sub-transforms of length Ns become length Ns*R.  tw is the N-entry root table;
// the roots of the Nh-point transform are its even entries.
template <typename T, int R>
__device__ __forceinline__ void stockham_stage(const C2<T>* __restrict__ src, C2<T>* __restrict__ dst, int Nh, int Ns,
                                               const C2<T>* __restrict__ tw, int N) {
    const int m = Nh / R;
    const int tstep = 2 * (Nh / (Ns * R));            // tw index step for exp(-2 pi i k / (Ns R))
    // j = q Ns + k: a shift and a mask while the sub-transform length is a power of two (every stage of a power-of-two block,
    // the leading stages of a mixed-radix one); integer division is ~25 instructions on this machine, more than the butterfly
    const bool pow2 = (Ns & (Ns - 1)) == 0;
    const int sh = __ffs(Ns) - 1;
    for (int j = threadIdx.x; j < m; j += STFT_TPB) {
        const int q = pow2 ? (j >> sh) : (j / Ns);
        const int k = j - q * Ns;
        C2<T> u[R];
#pragma unroll
        for (int t = 0; t < R; ++t) u[t] = src[j + t * m];
        // twiddles w^t, w = exp(-2 pi i k / (Ns R)): none in the first stage (k = 0); radix 4 fetches w alone and squares and
        // cubes it (a table read is an L2 round trip on the critical path of the stage, a complex product is four FMAs)
        if (Ns > 1) {
            if constexpr (R == 4) {
                const C2<T> w1 = tw[k * tstep], w2 = cmul(w1, w1), w3 = cmul(w2, w1);
                u[1] = cmul(u[1], w1);
                u[2] = cmul(u[2], w2);
                u[3] = cmul(u[3], w3);
            } else {
#pragma unroll
                for (int t = 1; t < R; ++t) u[t] = cmul(u[t], tw[t * k * tstep]);   // t k tstep <= (R-1)(Ns-1) N / (Ns R) < N: no wrap
            }
        }
        C2<T> v[R];
        if constexpr (R == 2) {
            v[0] = cadd(u[0], u[1]);
            v[1] = csub(u[0], u[1]);
        } else if constexpr (R == 4) {
            const C2<T> a = cadd(u[0], u[2]), b = csub(u[0], u[2]), c = cadd(u[1], u[3]), d = csub(u[1], u[3]);
            v[0] = cadd(a, c);
            v[2] = csub(a, c);
            v[1] = c2<T>(b.x + d.y, b.y - d.x);       // b - i d
            v[3] = c2<T>(b.x - d.y, b.y + d.x);       // b + i d
        } else {
            // direct DFT of prime length R with the roots w_R^q = tw[q N / R]
            const int rstep = N / R;
#pragma unroll
            for (int o = 0; o < R; ++o) {
                C2<T> acc = u[0];
#pragma unroll
                for (int t = 1; t < R; ++t) acc = cadd(acc, cmul(u[t], tw[((o * t) % R) * rstep]));
                v[o] = acc;
            }
        }
        const int base = q * Ns * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) dst[base + t * Ns] = v[t];
    }
}

// The same stage on ONE buffer: every thread reads the inputs of all its butterflies, the workgroup meets, the outputs
// go back to the same array.  Two barriers per stage instead of one, half the LDS: at 16 KB instead of 32 KB per
// 2048-point double-precision transform eight workgroups fit a CU instead of five -- and five instead of two beside
// the streaming pipeline's joint diagonalisation, which holds 80 KB of every CU while the next hop's transforms run.
template <typename T, int R, int MI = INPLACE_MAX_IT>
__device__ __forceinline__ void stockham_stage_inplace(C2<T>* __restrict__ buf, int Nh, int Ns, const C2<T>* __restrict__ tw) {
    // MI: butterflies per thread held in registers across the barrier.  The array below is sized by it, whatever the plan needs at run
    // time: with MI = 4 the float64 transforms carried 64 VGPRs of it (101 in all: five workgroups per CU) although a 2048-point
    // block needs ONE butterfly per thread and stage; the hot kernels are therefore instantiated for MI = 1 as well (plan.max_it).
    static_assert(R == 2 || R == 4, "in-place stages are radix 2 or 4");
    const int m = Nh / R;
    const int tstep = 2 * (Nh / (Ns * R));
    const int sh = __ffs(Ns) - 1;                      // Ns is a power of two here
    C2<T> v[MI][R];
#pragma unroll
    for (int it = 0; it < MI; ++it) {
        const int j = threadIdx.x + it * STFT_TPB;
        if (j >= m) break;
        const int k = j & (Ns - 1);
        C2<T> u[R];
#pragma unroll
        for (int t = 0; t < R; ++t) u[t] = buf[j + t * m];
        if (Ns > 1) {
            const C2<T> w1 = tw[k * tstep];
            u[1] = cmul(u[1], w1);
            if constexpr (R == 4) {
                const C2<T> w2 = cmul(w1, w1), w3 = cmul(w2, w1);
                u[2] = cmul(u[2], w2);
                u[3] = cmul(u[3], w3);
            }
        }
        if constexpr (R == 2) {
            v[it][0] = cadd(u[0], u[1]);
            v[it][1] = csub(u[0], u[1]);
        } else {
            const C2<T> a = cadd(u[0], u[2]), b = csub(u[0], u[2]), c = cadd(u[1], u[3]), d = csub(u[1], u[3]);
            v[it][0] = cadd(a, c);
            v[it][2] = csub(a, c);
            v[it][1] = c2<T>(b.x + d.y, b.y - d.x);
            v[it][3] = c2<T>(b.x - d.y, b.y + d.x);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MI; ++it) {
        const int j = threadIdx.x + it * STFT_TPB;
        if (j >= m) break;
        const int q = j >> sh, k = j & (Ns - 1);
        const int base = q * Ns * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) buf[base + t * Ns] = v[it][t];
    }
    __syncthreads();
}

// forward complex FFT of length plan.Nh on natural-order data in `a`; returns the buffer holding the result (`b` is not
// touched, and need not exist, when plan.inplace is set)
template <typename T, int MI = INPLACE_MAX_IT>
__device__ __forceinline__ C2<T>* fft_forward(const FftPlan& plan, C2<T>* a, C2<T>* b, const C2<T>* __restrict__ tw) {
    if (plan.inplace) {
        int Ns = 1;
        for (int s = 0; s < plan.nstages; ++s) {
            if (plan.radix[s] == 4) stockham_stage_inplace<T, 4, MI>(a, plan.Nh, Ns, tw);
            else stockham_stage_inplace<T, 2, MI>(a, plan.Nh, Ns, tw);
            Ns *= plan.radix[s];
        }
        return a;
    }
    int Ns = 1;
    C2<T>* src = a;
    C2<T>* dst = b;
    for (int s = 0; s < plan.nstages; ++s) {
        const int R = plan.radix[s];
        switch (R) {
            case 4: stockham_stage<T, 4>(src, dst, plan.Nh, Ns, tw, plan.N); break;
            case 2: stockham_stage<T, 2>(src, dst, plan.Nh, Ns, tw, plan.N); break;
            case 3: stockham_stage<T, 3>(src, dst, plan.Nh, Ns, tw, plan.N); break;
            case 5: stockham_stage<T, 5>(src, dst, plan.Nh, Ns, tw, plan.N); break;
            default: stockham_stage<T, 7>(src, dst, plan.Nh, Ns, tw, plan.N); break;
        }
        __syncthreads();
        Ns *= R;
        C2<T>* t = src; src = dst; dst = t;
    }
    return src;
}

template <typename T, int MI = INPLACE_MAX_IT>
__device__ __forceinline__ void rfft_from_lds(const FftPlan& plan, C2<T>* za, C2<T>* zb, C2<T>* __restrict__ out, long stride_k,
                                              const C2<T>* __restrict__ tw, int kgroup = 1);

template <typename T, int MI = INPLACE_MAX_IT>
__device__ __forceinline__ void stft_analysis_body(const FftPlan& plan, const T* __restrict__ xin, int in_len, int ring_off,
                                                   int use_win, C2<T>* __restrict__ out, long stride_k,
                                                   const C2<T>* __restrict__ tw, const T* __restrict__ win, int kgroup = 1) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    C2<T>* za = reinterpret_cast<C2<T>*>(smem_raw);
    const int N = plan.N, Nh = plan.Nh;
    C2<T>* zb = za + Nh;
    const int tid = threadIdx.x;
    for (int n = tid; n < Nh; n += STFT_TPB) {
        int i0 = 2 * n + ring_off, i1 = i0 + 1;
        if (i0 >= N) i0 -= N;
        if (i1 >= N) i1 -= N;
        T v0, v1;
        if (plan.debug & 2) {
            v0 = (T)i0;
            v1 = (T)i1;
        } else {
            v0 = (2 * n < in_len) ? xin[i0] : (T)0;
            v1 = (2 * n + 1 < in_len) ? xin[i1] : (T)0;
        }
        if (use_win) {
            v0 *= win[2 * n];
            v1 *= win[2 * n + 1];
        }
        za[n] = c2<T>(v0, v1);
    }
    __syncthreads();
    rfft_from_lds<T, MI>(plan, za, zb, out, stride_k, tw, kgroup);
}

// the real series x[2n], x[2n+1] sits in za[n] (and the workgroup has met): spectrum to out[k * stride_k], k <= N/2
// kgroup > 1: GROUPED bin-major layout [K / g][C][g] (round 4): bins k .. k + g - 1 of a channel are contiguous (g = 4: one 64-byte
// line of c128), the groups stride_k * g elements apart; the caller has offset `out` by c * g.  A transform then fills whole
// lines -- 34.7 million 16-byte pieces per chunk of cfg3, each in another DRAM page, were what its stores cost.
template <typename T, int MI>
__device__ __forceinline__ void rfft_from_lds(const FftPlan& plan, C2<T>* za, C2<T>* zb, C2<T>* __restrict__ out, long stride_k,
                                              const C2<T>* __restrict__ tw, int kgroup) {
    const int Nh = plan.Nh, tid = threadIdx.x;
    const C2<T>* z = (plan.debug & 4) ? za : fft_forward<T, MI>(plan, za, zb, tw);
    // even/odd split: X[k] = E[k] + e^{-2 pi i k/N} O[k]
    // (unrolling this loop and the input loop above so that all of a thread's loads go out together was tried in round 3 after the
    // order-16 kernel's slab loads: 65 -> 97 VGPRs and no change in the kernel's 437 us per chunk -- the transforms of a chunk move
    // 1.1 GB, half of it as 16-byte pieces of bin-major lines, in that time)
    for (int k = tid; k <= Nh; k += STFT_TPB) {
        const C2<T> a = z[k == Nh ? 0 : k];
        const C2<T> bq = z[k == 0 ? 0 : Nh - k];
        const C2<T> b = c2<T>(bq.x, -bq.y);                              // conj(Z[Nh-k])
        const C2<T> e = c2<T>((T)0.5 * (a.x + b.x), (T)0.5 * (a.y + b.y));
        const C2<T> dm = c2<T>((T)0.5 * (a.x - b.x), (T)0.5 * (a.y - b.y));
        const C2<T> o = c2<T>(dm.y, -dm.x);                              // -i * dm
        const C2<T> ow = cmul(o, tw[k]);                                 // tw[Nh] = -1
        const size_t oi = kgroup > 1 ? (size_t)(k / kgroup) * kgroup * stride_k + (k % kgroup) : (size_t)k * stride_k;
        if (!(plan.debug & 1) || e.x == (T)12345.678) out[oi] = c2<T>(e.x + ow.x, e.y + ow.y);
    }
}

// x is a ring: logical sample n of channel c lives at x[c*x_stride + (n + ring_off) mod N]; samples
// n >= in_len read as zero (zero padding, apvast.py:417: rfft(taps, N)); use_win = 0 skips the window
template <typename T>
__global__ void __launch_bounds__(STFT_TPB) stft_analysis_kernel(FftPlan plan, const T* __restrict__ x, long x_stride,
                                                                 int in_len, int ring_off, int use_win,
                                                                 C2<T>* __restrict__ spec, long stride_c, long stride_k,
                                                                 const C2<T>* __restrict__ tw,
                                                                 const T* __restrict__ win) {
    const int c = blockIdx.x;
    stft_analysis_body<T>(plan, x + (size_t)c * x_stride, in_len, ring_off, use_win, spec + (size_t)c * stride_c, stride_k, tw, win);
}

// several channel sets (full-length, windowed, one shared ring offset) in one launch: the streaming hop has seven
constexpr int STFT_MAX_JOBS = 8;
template <typename T>
struct StftJobs {
    const T* x[STFT_MAX_JOBS];
    C2<T>* spec[STFT_MAX_JOBS];
    long stride_c[STFT_MAX_JOBS], stride_k[STFT_MAX_JOBS];
    int ch0[STFT_MAX_JOBS + 1];                  // first workgroup of each job
    int n;
    // a whole chunk of hops in one launch (blockIdx.y = hop of the chunk): channel rows are x_stride samples apart, hop i reads
    // its block x_hop samples further on and writes its spectra spec_hop[j] elements further on.  One hop: x_stride = N, rest 0
    long x_stride, x_hop;
    long spec_hop[STFT_MAX_JOBS];
};
template <typename T, int MI>
__global__ void __launch_bounds__(STFT_TPB) stft_analysis_jobs_kernel(FftPlan plan, StftJobs<T> jobs, int ring_off,
                                                                      const C2<T>* __restrict__ tw,
                                                                      const T* __restrict__ win) {
    // Workgroups go to the eight XCDs round-robin, and a channel writes its bins 16 (8) bytes at a stride of a whole row of the
    // bin-major spectra: channel c and its neighbours fill the same 64-byte lines.  Workgroup b therefore takes the b/8-th
    // channel of the (b mod 8)-th eighth of the launch, so that one XCD's L2 sees neighbouring channels back to back and
    // writes whole lines.
    const int total = jobs.ch0[jobs.n], q8 = total >> 3, r8 = total & 7;
    const int g8 = (int)blockIdx.x & 7, i8 = (int)blockIdx.x >> 3;
    const int wg = g8 * q8 + min(g8, r8) + i8;
    int j = 0;
    while (j + 1 < jobs.n && wg >= jobs.ch0[j + 1]) ++j;
    const int c = wg - jobs.ch0[j];
    const size_t hop = blockIdx.y;
    // timing aid (APV_STFT_DEBUG & 64; the spectra come out in the wrong places): every channel writes its bins as ONE contiguous run
    // inside the set's region -- what a channel-major spectra layout would cost this kernel
    const bool cm = (plan.debug & 64) && jobs.stride_c[j] == 1;
    // a set whose channel stride AND bin stride exceed one is in the grouped layout, the channel stride being the group (see
    // rfft_from_lds): bin-major sets have stride_c = 1, channel-major ones stride_k = 1
    const int kgroup = (jobs.stride_c[j] > 1 && jobs.stride_k[j] > 1) ? (int)jobs.stride_c[j] : 1;
    stft_analysis_body<T, MI>(plan, jobs.x[j] + (size_t)c * jobs.x_stride + hop * jobs.x_hop, plan.N, ring_off, 1,
                          jobs.spec[j] + hop * jobs.spec_hop[j] + (cm ? (size_t)c * (plan.Nh + 1) : (size_t)c * jobs.stride_c[j]),
                          cm ? 1 : jobs.stride_k[j], tw, win, kgroup);
}

// (Four neighbouring channels per 1024-thread workgroup, so that every bin's four values go out as one 64-byte line instead of four
// 16-byte pieces, was tried for the chunk launches and lost: 470 us against 440 us per chunk of 16 hops at cfg3.  The transforms are
// bound by their LDS round trips and barriers, not by the partial-line writes, and sixteen waves in step hide less of them than four
// independent workgroups.)
template <typename T, int MI>
__global__ void __launch_bounds__(STFT_TPB) istft_ola_kernel(FftPlan plan, int H, const C2<T>* __restrict__ spec,
                                                             long stride_c, long stride_k, T* __restrict__ overlap,
                                                             T* __restrict__ out, const C2<T>* __restrict__ tw,
                                                             const T* __restrict__ win, int out_group, long out_gstride) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    C2<T>* za = reinterpret_cast<C2<T>*>(smem_raw);
    const int N = plan.N, Nh = plan.Nh;
    C2<T>* zb = za + Nh;
    const int c = blockIdx.x, tid = threadIdx.x;
    const C2<T>* X = spec + (size_t)c * stride_c;
    // Z[k] = E[k] + i O[k];  inverse transform as conj(FFT(conj(Z))) / Nh
    for (int k = tid; k < Nh; k += STFT_TPB) {
        C2<T> a = X[(size_t)k * stride_k];
        C2<T> bq = X[(size_t)(Nh - k) * stride_k];
        if (k == 0) {                                                   // irfft drops imag of DC and Nyquist
            a.y = 0;
            bq.y = 0;
        }
        const C2<T> b = c2<T>(bq.x, -bq.y);
        const C2<T> e = c2<T>((T)0.5 * (a.x + b.x), (T)0.5 * (a.y + b.y));
        const C2<T> dm = c2<T>((T)0.5 * (a.x - b.x), (T)0.5 * (a.y - b.y));
        const C2<T> wk = tw[k];
        const C2<T> o = cmul(dm, c2<T>(wk.x, -wk.y));                    // * e^{+2 pi i k/N}
        za[k] = c2<T>(e.x - o.y, -(e.y + o.x));                          // conj(E + i O)
    }
    __syncthreads();
    C2<T>* z = fft_forward<T, MI>(plan, za, zb, tw);
    const T scale = (T)1 / (T)Nh;
    T* ov = overlap + (size_t)c * N;
    T* zf = reinterpret_cast<T*>(z);
    // windowed new block, in place: x[2n] = Re z[n], x[2n+1] = -Im(conj-FFT)[n]
    for (int n = tid; n < Nh; n += STFT_TPB) {
        const C2<T> v = z[n];
        z[n] = c2<T>(v.x * scale * win[2 * n], -v.y * scale * win[2 * n + 1]);
    }
    __syncthreads();
    for (int n = tid; n < N; n += STFT_TPB) {
        const T old = (n < N - H) ? ov[n + H] : (T)0;
        zf[n] += old;
    }
    __syncthreads();
    for (int n = tid; n < N; n += STFT_TPB) ov[n] = zf[n];
    if (out != nullptr) {
        if (out_group > 0) {
            // sample-major emit: channels come in groups of out_group loudspeakers (one zone program and rank each) and a group is
            // written [H][out_group], the (hop, loudspeaker) array the reference's caller receives (apvast.py:498-504); groups lie
            // out_gstride elements apart (H out_group for one hop on its own; more when the hop is a slice of a longer signal)
            const int g = c / out_group, l = c - g * out_group;
            T* const o = out + (size_t)g * out_gstride + l;
            for (int n = tid; n < H; n += STFT_TPB) o[(size_t)n * out_group] = zf[n];
        } else {
            for (int n = tid; n < H; n += STFT_TPB) out[(size_t)c * H + n] = zf[n];
        }
    }
}

// ---- K1 by fast convolution ---------------------------------------------------------------------------------
//   y[n][c] = sum_p xh[P-1+n-p] rir[p][c], n < H                                     apvast.py:171-192 (lfilter)
// as one overlap-save segment of length F >= P - 1 + H: the history xh[0 .. P-1+H) zero-padded to F has the spectrum Xf
// (fir_input_spectra_kernel, one transform per input signal and hop), the impulse responses zero-padded to F have the
// spectra Hf (once, at set-up), and sample P - 1 + n of irfft(Xf Hf[c]) is y[n][c]: the first P - 1 samples of the
// circular convolution are the wrapped ones and are not used.  One workgroup per control-point channel: 2 (F/2 + 1)
// spectrum values in, an F/2-point complex transform in LDS, H samples out into the channel's response ring.  At
// P = 800, H = 1024 (F = 2048) this is 0.25 Gflop and 53 MB (float64) per hop where the direct form is 3.4 Gflop.
//
// Responses too long for one segment in LDS (F > 4096 doubles / 8192 floats) are UNIFORMLY PARTITIONED (n_part > 1): partitions of
// H taps, segments of F = 2 H samples.  With X_q the spectrum of the input samples [(t - q - 1) H, (t - q + 1) H) of hop t and
// H_q[c] the spectrum of taps [q H, (q + 1) H) of channel c zero-padded to 2 H,
//     y_t[c] = last H samples of irfft( sum_q X_q H_q[c] ),
// i.e. the same kernel with a sum over the partitions in front of the inverse transform.  The n_part input spectra of a hop are
// formed afresh from the input history every hop (2 n_part small transforms): there is no delay line to checkpoint.
constexpr int FIR_FFT_JOBS = 6;
template <typename T>
struct FirFftJobs {
    const C2<T>* Hf[FIR_FFT_JOBS];    // [C_j][F/2 + 1]
    const C2<T>* Xf[FIR_FFT_JOBS];    // [F/2 + 1], spectrum of the job's input history
    T* resp[FIR_FFT_JOBS];            // [C_j][row_stride]: ring of N samples (one hop) or linear buffer (a chunk of hops)
    int ch0[FIR_FFT_JOBS + 1];        // first workgroup of each job
    int n;
    // where the H new samples of hop blockIdx.y go: (pos0 + hop H + i) mod row_stride of the channel's row; the input spectra of
    // hop i are x_hop_stride elements behind those of hop 0.  One hop: row_stride = N, pos0 = (N - H + ring offset) mod N
    int row_stride, pos0, n_hops;
    long x_hop_stride;
    // partitioned convolution: n_part partitions; partition q of job j's responses lies h_part_stride[j] elements behind partition
    // q - 1, of its input spectra x_part_stride elements; `skip` = first valid sample of the circular convolution (P - 1, or H when
    // partitioned).  One segment: n_part = 1
    int n_part, skip;
    long h_part_stride[FIR_FFT_JOBS], x_part_stride;
    // optional passenger (whole-signal path, where the hop's input spectra exist before its input update): the input
    // update of the hop -- new histories [old[H:], hop, zeros(pad)], hop appended to the input-block rings -- done by
    // upd_wgs extra workgroups per signal at the end of the grid instead of a launch of its own in front of this one
    int upd_wgs, pad;
    const T* old_hist[2];
    T* new_hist[2];
    const T* xin;                     // pinned host [2][H]
    T* inblk;                         // [2][N] rings
};

// blockIdx.y = partition q (one segment: gridDim.y = 1, part_step = 0): its segment starts part_step samples EARLIER in the history
// than that of partition q - 1, and its spectra go behind those of partition q - 1
template <typename T>
__global__ void __launch_bounds__(STFT_TPB) fir_input_spectra_kernel(FftPlan plan, const T* __restrict__ x0,
                                                                     const T* __restrict__ x1, int in_len, int part_step,
                                                                     C2<T>* __restrict__ spec, const C2<T>* __restrict__ tw) {
    const int q = blockIdx.y, nq = gridDim.y;
    const T* x = (blockIdx.x ? x1 : x0) + (size_t)(nq - 1 - q) * part_step;
    stft_analysis_body<T>(plan, x, in_len, 0, 0, spec + ((size_t)q * 2 + blockIdx.x) * (plan.Nh + 1), 1, tw, nullptr);
}

// The same spectra for a whole chunk of hops in one launch (blockIdx.x = signal, blockIdx.y = hop of the chunk), before any
// of them has been through the input update: the history of hop i is samples i H ... i H + P - 1 + H of the stream
// [last P - 1 samples of the history at the start of the chunk | hop 0 | hop 1 | ...], the hops read from the pinned
// host staging `pin` [hops][2][H].  Values and arithmetic are those of fir_input_spectra_kernel: the spectra are the same
// bit for bit; what changes is that no hop of the whole-signal path waits for a two-workgroup launch of its own.
template <typename T>
__global__ void __launch_bounds__(STFT_TPB) fir_chunk_spectra_kernel(FftPlan plan, int P, int H, const T* __restrict__ hist0,
                                                                     const T* __restrict__ hist1, const T* __restrict__ pin,
                                                                     C2<T>* __restrict__ spec, const C2<T>* __restrict__ tw) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    C2<T>* za = reinterpret_cast<C2<T>*>(smem_raw);
    C2<T>* zb = za + plan.Nh;
    const int g = blockIdx.x, hop = blockIdx.y, keep = P - 1, len = keep + H;
    const T* __restrict__ hist = g ? hist1 : hist0;                 // [P - 1 + H]: its last P - 1 samples precede the chunk
    auto sample = [&](int n) -> T {                                   // n-th sample of this hop's history
        if (n >= len) return (T)0;
        const int m = hop * H + n;                                    // position in the stream described above
        if (m < keep) return hist[H + m];
        const int q = (m - keep) / H, t = (m - keep) - q * H;
        return pin[((size_t)q * 2 + g) * H + t];
    };
    for (int n = threadIdx.x; n < plan.Nh; n += STFT_TPB) za[n] = c2<T>(sample(2 * n), sample(2 * n + 1));
    __syncthreads();
    rfft_from_lds<T>(plan, za, zb, spec + ((size_t)hop * 2 + g) * (plan.Nh + 1), 1, tw);
}

// (Four hops of a channel per 1024-thread workgroup, so that the channel's response spectrum comes from L2 once per four hops, was
// tried for the chunk launches and lost: 430 us against 300 us per chunk of 16 hops at cfg3 -- sixteen waves meeting at the same
// barriers hide less than eight independent workgroups of four.)
template <typename T, int MI>
__global__ void __launch_bounds__(STFT_TPB) fir_fft_kernel(FftPlan plan, FirFftJobs<T> jobs, int P, int H, int N, int ring_off,
                                                           const C2<T>* __restrict__ tw) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    using Z = C2<T>;
    Z* za = reinterpret_cast<Z*>(smem_raw);
    const int Fh = plan.Nh;
    Z* zb = za + Fh;
    const int wg = blockIdx.x, tid = threadIdx.x;
    if (wg >= jobs.ch0[jobs.n]) {                                   // the passenger: same work as input_update_kernel
        const int u = wg - jobs.ch0[jobs.n], g = u / jobs.upd_wgs, i = (u - g * jobs.upd_wgs) * STFT_TPB + tid;
        const T* x = jobs.xin + (size_t)g * H;
        const int keep = P - 1;
        if (i < keep) jobs.new_hist[g][i] = jobs.old_hist[g][i + H];
        else if (i < keep + H) jobs.new_hist[g][i] = x[i - keep];
        else if (i < keep + H + jobs.pad) jobs.new_hist[g][i] = (T)0;
        if (i < H) jobs.inblk[(size_t)g * N + (N - H + i + ring_off) % N] = x[i];
        return;
    }
    int j = 0;
    while (j + 1 < jobs.n && wg >= jobs.ch0[j + 1]) ++j;
    const int c = wg - jobs.ch0[j];
    const Z* __restrict__ Hc = jobs.Hf[j] + (size_t)c * (Fh + 1);
    const int hop = blockIdx.y;
    const Z* __restrict__ X = jobs.Xf[j] + (size_t)hop * jobs.x_hop_stride;
    // product spectrum (summed over the partitions), packed for the half-length inverse transform exactly as in istft_ola_kernel
    const int n_part = jobs.n_part;
    const long hps = jobs.h_part_stride[j], xps = jobs.x_part_stride;
    auto pack = [&](int k, Z a, Z bq, Z wk) {
        for (int q = 1; q < n_part; ++q) {
            a = cadd(a, cmul(X[q * xps + k], Hc[q * hps + k]));
            bq = cadd(bq, cmul(X[q * xps + Fh - k], Hc[q * hps + Fh - k]));
        }
        if (k == 0) {
            a.y = 0;
            bq.y = 0;
        }
        const Z b = c2<T>(bq.x, -bq.y);
        const Z e = c2<T>((T)0.5 * (a.x + b.x), (T)0.5 * (a.y + b.y));
        const Z dm = c2<T>((T)0.5 * (a.x - b.x), (T)0.5 * (a.y - b.y));
        const Z o = cmul(dm, c2<T>(wk.x, -wk.y));
        za[k] = c2<T>(e.x - o.y, -(e.y + o.x));
    };
    // two bins and their mirrors per thread and pass, all ten loads issued before the first product (a one-bin loop made every
    // pass a memory round trip of its own: the response spectra come from the Infinity Cache)
    for (int k0 = tid; k0 < Fh; k0 += 2 * STFT_TPB) {
        Z xa[2], ha[2], xb[2], hb[2], wk[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + u * STFT_TPB, kk = k < Fh ? k : 0;
            xa[u] = X[kk]; ha[u] = Hc[kk]; xb[u] = X[Fh - kk]; hb[u] = Hc[Fh - kk]; wk[u] = tw[kk];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = k0 + u * STFT_TPB;
            if (k < Fh) pack(k, cmul(xa[u], ha[u]), cmul(xb[u], hb[u]), wk[u]);
        }
    }
    __syncthreads();
    const Z* z = fft_forward<T, MI>(plan, za, zb, tw);
    const T scale = (T)1 / (T)Fh;
    T* __restrict__ dst = jobs.resp[j] + (size_t)c * jobs.row_stride;
    const int p0 = jobs.pos0 + hop * H;
    for (int i = tid; i < H; i += STFT_TPB) {
        const int n = jobs.skip + i;
        const Z v = z[n >> 1];
        dst[(p0 + i) % jobs.row_stride] = ((n & 1) ? -v.y : v.x) * scale;
    }
}

bool make_plan(int N, FftPlan* plan, std::string* why) {
    if (N < 4 || N > STFT_MAX_N || (N & 1)) {
        if (why) *why = "STFT block size must be even and in [4, 8192]";
        return false;
    }
    plan->N = N;
    plan->Nh = N / 2;
    plan->nstages = 0;
    int rem = N / 2;
    const int radices[] = {4, 2, 3, 5, 7};
    for (int r : radices)
        while (rem % r == 0 && rem > 1) {
            if (plan->nstages >= MAX_STAGES) break;
            plan->radix[plan->nstages++] = r;
            rem /= r;
        }
    if (rem != 1) {
        if (why) *why = "STFT block size / 2 must factor into 2, 3, 5 and 7";
        return false;
    }
    static const bool pingpong = getenv("APV_FFT_PINGPONG") != nullptr;      // A/B switch: two-buffer stages everywhere
    plan->inplace = pingpong ? 0 : 1;
    plan->max_it = 1;
    // Timing aids of the probes: they make every transform WRONG.  A stray APV_STFT_DEBUG in somebody's environment must not
    // corrupt a stream silently: the bits are honoured only together with APV_STFT_DEBUG_PROBE=1, which the probe
    // scripts set; alone, the variable refuses the plan -- no stream can be created with it.
    plan->debug = 0;
    if (getenv("APV_STFT_DEBUG") && atoi(getenv("APV_STFT_DEBUG")) != 0) {
        if (!getenv("APV_STFT_DEBUG_PROBE")) {
            if (why) *why = "APV_STFT_DEBUG is set (timing aids that make every transform wrong) without APV_STFT_DEBUG_PROBE=1: unset it";
            return false;
        }
        plan->debug = atoi(getenv("APV_STFT_DEBUG"));
        static bool warned = false;
        if (!warned) fprintf(stderr, "[apv] APV_STFT_DEBUG=%d: the transforms of this process are timing aids, their results are WRONG\n", plan->debug);
        warned = true;
    }
    for (int s = 0; s < plan->nstages; ++s) {
        const int r = plan->radix[s];
        if ((r != 2 && r != 4) || plan->Nh / r > INPLACE_MAX_IT * STFT_TPB) plan->inplace = 0;
        const int it = (plan->Nh / r + STFT_TPB - 1) / STFT_TPB;
        if (it > plan->max_it) plan->max_it = it;
    }
    static const bool mi4 = getenv("APV_FFT_MI4") != nullptr;                // A/B switch: the four-butterfly instantiations everywhere
    if (!plan->inplace || mi4) plan->max_it = INPLACE_MAX_IT;
    return true;
}

// LDS bytes of one transform
template <typename T>
size_t plan_lds(const FftPlan& plan) { return sizeof(C2<T>) * (plan.inplace ? 1 : 2) * plan.Nh; }

template <typename T>
hipError_t launch_analysis(int N, int n_ch, const void* x, long x_stride, int in_len, int ring_off, int use_win,
                           void* spec, long stride_c, long stride_k, hipStream_t s, std::string* why) {
    FftPlan plan;
    if (!make_plan(N, &plan, why)) return hipErrorInvalidValue;
    if (n_ch <= 0) return hipSuccess;
    Tables<T> t;
    hipError_t e = get_tables<T>(N, &t);
    if (e != hipSuccess) return e;
    const size_t lds = plan_lds<T>(plan);
    int off = ring_off % N;
    if (off < 0) off += N;
    hipLaunchKernelGGL(stft_analysis_kernel<T>, dim3(n_ch), dim3(STFT_TPB), lds, s, plan, (const T*)x, x_stride, in_len,
                       off, use_win, (C2<T>*)spec, stride_c, stride_k, t.tw, t.win);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_synthesis(int N, int H, int n_ch, const void* spec, long stride_c, long stride_k, void* overlap,
                            void* out, hipStream_t s, std::string* why, int out_group = 0, long out_gstride = 0) {
    FftPlan plan;
    if (!make_plan(N, &plan, why)) return hipErrorInvalidValue;
    if (H <= 0 || H > N) {
        if (why) *why = "hop size out of range";
        return hipErrorInvalidValue;
    }
    if (n_ch <= 0) return hipSuccess;
    Tables<T> t;
    hipError_t e = get_tables<T>(N, &t);
    if (e != hipSuccess) return e;
    const size_t lds = plan_lds<T>(plan);
    hipLaunchKernelGGL((plan.max_it == 1 ? istft_ola_kernel<T, 1> : istft_ola_kernel<T, INPLACE_MAX_IT>), dim3(n_ch), dim3(STFT_TPB), lds, s, plan, H, (const C2<T>*)spec, stride_c,
                       stride_k, (T*)overlap, (T*)out, t.tw, t.win, (out_group > 0 && n_ch % out_group == 0) ? out_group : 0,
                       out_gstride > 0 ? out_gstride : (long)H * out_group);
    return hipGetLastError();
}

}  // namespace

namespace {
template <typename T>
hipError_t launch_analysis_jobs(const FftPlan& plan, int n_jobs, const void* const* x, const int* n_ch, void* const* spec,
                                const long* stride_c, const long* stride_k, int ring_off, hipStream_t s, long x_stride = 0,
                                long x_hop = 0, const long* spec_hop = nullptr, int n_hops = 1) {
    Tables<T> t;
    hipError_t e = get_tables<T>(plan.N, &t);
    if (e != hipSuccess) return e;
    StftJobs<T> jobs{};
    int total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        jobs.x[j] = (const T*)x[j];
        jobs.spec[j] = (C2<T>*)spec[j];
        jobs.stride_c[j] = stride_c[j];
        jobs.stride_k[j] = stride_k[j];
        jobs.spec_hop[j] = spec_hop ? spec_hop[j] : 0;
        jobs.ch0[j] = total;
        total += n_ch[j];
    }
    jobs.ch0[n_jobs] = total;
    jobs.n = n_jobs;
    jobs.x_stride = x_stride > 0 ? x_stride : plan.N;
    jobs.x_hop = x_hop;
    if (total <= 0 || n_hops <= 0) return hipSuccess;
    int off = ring_off % plan.N;
    if (off < 0) off += plan.N;
    hipLaunchKernelGGL((plan.max_it == 1 ? stft_analysis_jobs_kernel<T, 1> : stft_analysis_jobs_kernel<T, INPLACE_MAX_IT>), dim3(total, n_hops), dim3(STFT_TPB), plan_lds<T>(plan), s, plan, jobs, off,
                       t.tw, t.win);
    return hipGetLastError();
}

}  // namespace

hipError_t apv_launch_stft_analysis_jobs(int f64, int N, int n_jobs, const void* const* x, const int* n_ch, void* const* spec,
                                         const long* stride_c, const long* stride_k, int ring_off, hipStream_t s,
                                         std::string* why) {
    FftPlan plan;
    if (!make_plan(N, &plan, why)) return hipErrorInvalidValue;
    if (n_jobs < 1 || n_jobs > STFT_MAX_JOBS) return hipErrorInvalidValue;
    return f64 ? launch_analysis_jobs<double>(plan, n_jobs, x, n_ch, spec, stride_c, stride_k, ring_off, s)
               : launch_analysis_jobs<float>(plan, n_jobs, x, n_ch, spec, stride_c, stride_k, ring_off, s);
}

// The analysis transforms of a whole chunk of hops in one launch (whole-signal path): the blocks of hop i start x_hop * i samples
// into rows that are x_stride samples long (linear buffers: no ring offset), its spectra go spec_hop[j] * i elements further on
hipError_t apv_launch_stft_analysis_chunk(int f64, int N, int n_jobs, const void* const* x, const int* n_ch, void* const* spec,
                                          const long* stride_c, const long* stride_k, long x_stride, long x_hop, const long* spec_hop,
                                          int n_hops, hipStream_t s, std::string* why) {
    FftPlan plan;
    if (!make_plan(N, &plan, why)) return hipErrorInvalidValue;
    if (n_jobs < 1 || n_jobs > STFT_MAX_JOBS || n_hops < 1 || n_hops > 65535) return hipErrorInvalidValue;
    return f64 ? launch_analysis_jobs<double>(plan, n_jobs, x, n_ch, spec, stride_c, stride_k, 0, s, x_stride, x_hop, spec_hop, n_hops)
               : launch_analysis_jobs<float>(plan, n_jobs, x, n_ch, spec, stride_c, stride_k, 0, s, x_stride, x_hop, spec_hop, n_hops);
}

// build (or find) the twiddle / window tables now, so that no allocation happens on the per-hop path
hipError_t apv_stft_prepare(int N, int f64) {
    if (f64) { Tables<double> t; return get_tables<double>(N, &t); }
    Tables<float> t;
    return get_tables<float>(N, &t);
}

bool apv_stft_size_ok(int N, std::string* why) {
    FftPlan plan;
    return make_plan(N, &plan, why);
}

hipError_t apv_launch_analysis(int f64, int N, int n_ch, const void* x, long x_stride, int in_len, int ring_off,
                               int use_win, void* spec, long stride_c, long stride_k, hipStream_t s, std::string* why) {
    return f64 ? launch_analysis<double>(N, n_ch, x, x_stride, in_len, ring_off, use_win, spec, stride_c, stride_k, s, why)
               : launch_analysis<float>(N, n_ch, x, x_stride, in_len, ring_off, use_win, spec, stride_c, stride_k, s, why);
}

hipError_t apv_launch_synthesis(int f64, int N, int H, int n_ch, const void* spec, long stride_c, long stride_k,
                                void* overlap, void* out, hipStream_t s, std::string* why, int out_group, long out_gstride) {
    return f64 ? launch_synthesis<double>(N, H, n_ch, spec, stride_c, stride_k, overlap, out, s, why, out_group, out_gstride)
               : launch_synthesis<float>(N, H, n_ch, spec, stride_c, stride_k, overlap, out, s, why, out_group, out_gstride);
}

hipError_t apv_launch_stft_analysis_strided(int N, int n_ch, const float* x, int ring_off, float2* spec,
                                            long stride_c, long stride_k, hipStream_t s, std::string* why) {
    return launch_analysis<float>(N, n_ch, x, N, N, ring_off, 1, spec, stride_c, stride_k, s, why);
}

hipError_t apv_launch_istft_ola_strided(int N, int H, int n_ch, const float2* spec, long stride_c, long stride_k,
                                        float* overlap, float* out, hipStream_t s, std::string* why) {
    return launch_synthesis<float>(N, H, n_ch, spec, stride_c, stride_k, overlap, out, s, why);
}

hipError_t apv_launch_stft_analysis(int N, int n_ch, const float* x, float2* spec, hipStream_t s, std::string* why) {
    return apv_launch_stft_analysis_strided(N, n_ch, x, 0, spec, N / 2 + 1, 1, s, why);
}

hipError_t apv_launch_istft_ola(int N, int H, int n_ch, const float2* spec, float* overlap, float* out,
                                hipStream_t s, std::string* why) {
    return apv_launch_istft_ola_strided(N, H, n_ch, spec, N / 2 + 1, 1, overlap, out, s, why);
}

// ---- K1 by fast convolution: launchers ------------------------------------------------------------------------------
// Segment length for (P, H): the power of two >= P - 1 + H, or 0 when the direct form is the better choice (short
// responses) or the segment does not fit the transform in LDS (64 KB: 4096 doubles or 8192 floats).
int apv_fir_fft_size(int f64, int P, int H) {
    if (P < 64) return 0;
    int F = 64;
    while (F < P - 1 + H) F *= 2;
    return F <= (f64 ? 4096 : 8192) ? F : 0;
}

// Responses too long for one segment: partitions of H taps in segments of 2 H samples, when such a segment transforms in LDS and the
// history reaches back at least one hop (P - 1 >= H: otherwise the response is short and H is what is long; the direct form stays)
int apv_fir_partitions(int f64, int P, int H) {
    if (apv_fir_fft_size(f64, P, H) != 0 || P < 64 || P - 1 < H) return 0;
    const int F = 2 * H;
    if (F > (f64 ? 4096 : 8192) || !apv_stft_size_ok(F, nullptr)) return 0;
    FftPlan plan;
    if (!make_plan(F, &plan, nullptr)) return 0;
    return (P + H - 1) / H;
}

// spectra of n_ch impulse responses held channel-major, x [n_ch][P] -> Hf [n_ch][F/2 + 1]
hipError_t apv_launch_fir_spectra(int f64, int F, int n_ch, const void* x, int P, void* Hf, hipStream_t s, std::string* why) {
    return f64 ? launch_analysis<double>(F, n_ch, x, P, P, 0, 0, Hf, F / 2 + 1, 1, s, why)
               : launch_analysis<float>(F, n_ch, x, P, P, 0, 0, Hf, F / 2 + 1, 1, s, why);
}
// one partition: `taps` taps of every channel starting at x (rows x_stride samples apart), zero-padded to F
hipError_t apv_launch_fir_spectra_part(int f64, int F, int n_ch, const void* x, long x_stride, int taps, void* Hf, hipStream_t s,
                                       std::string* why) {
    return f64 ? launch_analysis<double>(F, n_ch, x, x_stride, taps, 0, 0, Hf, F / 2 + 1, 1, s, why)
               : launch_analysis<float>(F, n_ch, x, x_stride, taps, 0, 0, Hf, F / 2 + 1, 1, s, why);
}

namespace {
template <typename T>
hipError_t launch_fir_input_spectra(int F, const void* x0, const void* x1, int in_len, void* Xf, hipStream_t s, int n_part = 1,
                                    int part_step = 0) {
    FftPlan plan;
    if (!make_plan(F, &plan, nullptr)) return hipErrorInvalidValue;
    Tables<T> t;
    hipError_t e = get_tables<T>(F, &t);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fir_input_spectra_kernel<T>, dim3(2, n_part), dim3(STFT_TPB), plan_lds<T>(plan), s, plan, (const T*)x0,
                       (const T*)x1, in_len, part_step, (C2<T>*)Xf, t.tw);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_fir_chunk_spectra(int F, int P, int H, int n_hops, const void* hist0, const void* hist1, const void* pin, void* Xf,
                                    hipStream_t s) {
    FftPlan plan;
    if (!make_plan(F, &plan, nullptr) || P - 1 + H > F || n_hops < 1) return hipErrorInvalidValue;
    Tables<T> t;
    hipError_t e = get_tables<T>(F, &t);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fir_chunk_spectra_kernel<T>, dim3(2, n_hops), dim3(STFT_TPB), plan_lds<T>(plan), s, plan, P, H, (const T*)hist0,
                       (const T*)hist1, (const T*)pin, (C2<T>*)Xf, t.tw);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_fir_fft_jobs(int F, int n_jobs, const void* const* Hf, const void* const* Xf, void* const* resp, const int* n_ch,
                               int P, int H, int N, int ring_off, const ApvInputUpdate* upd, hipStream_t s, int row_stride = 0,
                               int pos0 = 0, long x_hop_stride = 0, int n_hops = 1, int n_part = 1) {
    FftPlan plan;
    if (!make_plan(F, &plan, nullptr) || n_jobs < 1 || n_jobs > FIR_FFT_JOBS || n_part < 1) return hipErrorInvalidValue;
    if (n_part == 1 ? (P - 1 + H > F) : (F != 2 * H || (long)n_part * H < P)) return hipErrorInvalidValue;
    Tables<T> t;
    hipError_t e = get_tables<T>(F, &t);
    if (e != hipSuccess) return e;
    FirFftJobs<T> jobs{};
    int total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        jobs.Hf[j] = (const C2<T>*)Hf[j];
        jobs.Xf[j] = (const C2<T>*)Xf[j];
        jobs.resp[j] = (T*)resp[j];
        jobs.h_part_stride[j] = (long)n_ch[j] * (F / 2 + 1);        // responses partition-major: Hf[q][c][F/2 + 1]
        jobs.ch0[j] = total;
        total += n_ch[j];
    }
    jobs.ch0[n_jobs] = total;
    jobs.n = n_jobs;
    jobs.n_part = n_part;
    jobs.x_part_stride = 2L * (F / 2 + 1);                          // input spectra [q][signal][F/2 + 1]
    jobs.skip = n_part == 1 ? P - 1 : F - H;
    if (total <= 0) return hipSuccess;
    int off = ring_off % N;
    if (off < 0) off += N;
    if (row_stride > 0) {                      // a chunk of hops into linear buffers
        jobs.row_stride = row_stride;
        jobs.pos0 = pos0;
        jobs.x_hop_stride = x_hop_stride;
        jobs.n_hops = n_hops;
        if (upd || n_hops < 1 || n_hops > 65535 || pos0 + (long)n_hops * H > row_stride) return hipErrorInvalidValue;
    } else {
        jobs.row_stride = N;
        jobs.pos0 = (N - H + off) % N;
        jobs.x_hop_stride = 0;
        n_hops = 1;
        jobs.n_hops = 1;
    }
    if (upd) {
        jobs.upd_wgs = (P - 1 + H + upd->pad + STFT_TPB - 1) / STFT_TPB;
        jobs.pad = upd->pad;
        for (int g = 0; g < 2; ++g) {
            jobs.old_hist[g] = (const T*)upd->old_hist[g];
            jobs.new_hist[g] = (T*)upd->new_hist[g];
        }
        jobs.xin = (const T*)upd->xin;
        jobs.inblk = (T*)upd->inblk;
        total += 2 * jobs.upd_wgs;
    }
    hipLaunchKernelGGL((plan.max_it == 1 ? fir_fft_kernel<T, 1> : fir_fft_kernel<T, INPLACE_MAX_IT>), dim3(total, n_hops), dim3(STFT_TPB), plan_lds<T>(plan), s, plan, jobs, P, H, N, off, t.tw);
    return hipGetLastError();
}
}  // namespace

// K1 of a whole chunk of hops in one launch: Xf[j] points at the input spectrum of the chunk's first hop (the following hops'
// x_hop_stride complex elements apart), resp[j] at rows of row_stride samples whose positions pos0 + i H ... take hop i
hipError_t apv_launch_fir_fft_chunk(int f64, int F, int n_jobs, const void* const* Hf, const void* const* Xf, long x_hop_stride,
                                    void* const* resp, const int* n_ch, int P, int H, int row_stride, int pos0, int n_hops,
                                    hipStream_t s) {
    return f64 ? launch_fir_fft_jobs<double>(F, n_jobs, Hf, Xf, resp, n_ch, P, H, row_stride, 0, nullptr, s, row_stride, pos0, x_hop_stride, n_hops)
               : launch_fir_fft_jobs<float>(F, n_jobs, Hf, Xf, resp, n_ch, P, H, row_stride, 0, nullptr, s, row_stride, pos0, x_hop_stride, n_hops);
}

// Xf [2][F/2 + 1] = spectra of the two input histories x0, x1 (in_len = P - 1 + H samples each, zero-padded to F)
hipError_t apv_launch_fir_input_spectra(int f64, int F, const void* x0, const void* x1, int in_len, void* Xf, hipStream_t s) {
    return f64 ? launch_fir_input_spectra<double>(F, x0, x1, in_len, Xf, s) : launch_fir_input_spectra<float>(F, x0, x1, in_len, Xf, s);
}

// partitioned K1: Xf [n_part][2][F/2 + 1], partition q = spectrum of the F = 2 H history samples that end q H samples before the
// history's end; x0, x1: histories of (n_part + 1) H samples
hipError_t apv_launch_fir_input_spectra_parts(int f64, int F, int n_part, const void* x0, const void* x1, void* Xf, hipStream_t s) {
    if (n_part < 1 || n_part > 65535) return hipErrorInvalidValue;
    return f64 ? launch_fir_input_spectra<double>(F, x0, x1, F, Xf, s, n_part, F / 2) : launch_fir_input_spectra<float>(F, x0, x1, F, Xf, s, n_part, F / 2);
}

// Xf [n_hops][2][F/2 + 1] for the hops of a staged chunk (pin [n_hops][2][H], host-pinned), from the histories as they are
// BEFORE the first of them is processed
hipError_t apv_launch_fir_chunk_spectra(int f64, int F, int P, int H, int n_hops, const void* hist0, const void* hist1,
                                        const void* pin, void* Xf, hipStream_t s) {
    return f64 ? launch_fir_chunk_spectra<double>(F, P, H, n_hops, hist0, hist1, pin, Xf, s)
               : launch_fir_chunk_spectra<float>(F, P, H, n_hops, hist0, hist1, pin, Xf, s);
}

hipError_t apv_launch_fir_fft_jobs(int f64, int F, int n_jobs, const void* const* Hf, const void* const* Xf, void* const* resp,
                                   const int* n_ch, int P, int H, int N, int ring_off, const ApvInputUpdate* upd, hipStream_t s, int n_part) {
    return f64 ? launch_fir_fft_jobs<double>(F, n_jobs, Hf, Xf, resp, n_ch, P, H, N, ring_off, upd, s, 0, 0, 0, 1, n_part)
               : launch_fir_fft_jobs<float>(F, n_jobs, Hf, Xf, resp, n_ch, P, H, N, ring_off, upd, s, 0, 0, 0, 1, n_part);
}
