// Internal declarations shared by the HIP translation units of libapvast_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/apvast_hip.h"

struct GevdParams {
    int n;            // GEVD order (= L in subband mode)
    int M;            // control points per zone (fused path only)
    int K;            // problems in this launch
    int nV;
    int ranks[APV_MAX_RANKS];
    double mu;
    double reg_dark;
    double reg_bright;
    int reg_mode;
    int max_sweeps;
    double sweep_tol2; // Jacobi stop threshold on off^2/||C||_F^2 seen during a sweep; 0 = per-dtype default
    int out_c128;
    // fused input (c64)
    const float2* XB;
    const float2* XD;
    const float2* d;
    // explicit input (complex of the compute dtype), row-major n x n
    const void* RB;
    const void* RD;
    const void* r;
    // outputs
    void* w;          // [K][nV][n]  c64 | c128
    void* lam;        // [K][n]      f32 | f64   (may be null)
    int32_t* status;  // [K]                      (may be null)
    void* U;          // [K][n][n]   complex of compute dtype, sorted columns (may be null)
    void* Lspill;     // [K][n][n]   complex of compute dtype (SPILL instances only)
};

struct apv_handle {
    apv_config cfg;
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    // workspaces
    void* d_XB;       // staging for the host-pointer entry points
    void* d_XD;
    void* d_d;
    void* d_w;
    void* d_lam;
    int32_t* d_status;
    void* d_Lspill;
    size_t lspill_bytes;
    void* comm;       // ncclComm_t
    int comm_rank, comm_world;
    std::string err;
};

// kernels_gevd.hip
hipError_t apv_launch_gevd(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s, std::string* why);
size_t apv_gevd_spill_bytes(int n, int K, int compute_dtype);

// kernels_gevd16.hip (order-16 fast path; hipErrorNotSupported when the problem does not qualify)
hipError_t apv_launch_gevd16(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s);

// kernels_corr.hip
hipError_t apv_launch_corr(int compute_dtype, int K, int M, int L, const float2* XB, const float2* XD,
                           const float2* d, void* RB, void* RD, void* r, hipStream_t s);

// kernels_stft.hip
hipError_t apv_launch_stft_analysis(int N, int n_ch, const float* x, float2* spec, hipStream_t s, std::string* why);
hipError_t apv_launch_istft_ola(int N, int H, int n_ch, const float2* spec, float* overlap, float* out,
                                hipStream_t s, std::string* why);
