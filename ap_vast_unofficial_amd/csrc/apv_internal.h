// Internal declarations shared by the HIP translation units of libapvast_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

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
    int debug_stop;    // profiling aid: return after stage N (1 = correlate, 2 = Cholesky, 3 = whitening); 0 = run everything
    double sweep_tol2; // Jacobi stop threshold on off^2/||C||_F^2 seen during a sweep; 0 = per-dtype default
    int out_c128;
    // fused input: c64, or c128 when x_c128 is set (the float64 streaming front-end)
    int x_c128;
    // fused input in the grouped layout [K / x_group][M n][x_group] (0 or 1: bin-major [K][M][n]); only the order-16 float64 kernel
    // on c128 slabs reads it (kernels_gevd16m.hip), every other kernel refuses a launch with x_group > 1
    int x_group;
    const void* XB;
    const void* XD;
    const void* d;
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
    // optional second zone program in the same launch (blockIdx.y == 1), fused path only
    int n_zones;
    const void* XB1;
    const void* XD1;
    const void* d1;
    void* w1;
    void* lam1;
    int32_t* status1;
    // streaming: 1 = this launch shares the chip with the transforms of the NEXT chunk of hops, which are the longer chain
    // (chunked whole-signal path): the waves keep the default issue priority instead of raising theirs
    int yield_issue;
    // diagnostic builds only (tools/probes/stage_stamps.py): s_memtime at the stage boundaries, [zones][K][16]; null in normal use
    unsigned long long* stamps;
    // several hops of a chunk in ONE launch (chunked whole-signal path; blockIdx.z = hop): the same K bins for n_hops consecutive spectra
    // sets, byte strides from hop to hop of the slabs (XB, XD and their second-zone twins), the targets, and the three outputs.  Only
    // the order-16 kernels on fused slabs take it (apv_gevd16m_takes_hops); 0 or 1: a single hop, strides unused
    int n_hops;
    size_t hop_X, hop_d, hop_w, hop_lam, hop_status;
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
    void* d_Rscratch;  // [2][K][L][L] + [K][L] c64: MFMA correlation output of the split n in {32, 64} f32 update
    size_t rscratch_bytes;
    unsigned long long* d_stamps;   // apv_debug_set_stamps: stage-stamp buffer of the diagnostic kernel instantiation (caller's)
    struct apv_stream* st;   // streaming state (apv_stream_init), owned
    struct apv_bb* bb;       // broadband streaming state (apv_bb_init), owned
    std::vector<int> bb_rank_list;   // apv_bb_set_rank_list: ranks of the next apv_bb_init (empty = 1..V)
    void* gl_ws;             // workspace + captured sweep graph of apv_gevd_large, owned
    double gl_tol2;          // > 0: stop threshold of apv_gevd_large's sweeps for the next call (the complex path asks for accurate eigenVECTORS)
    int gl_lead_rank;        // > 0: the next apv_gevd_large call needs the leading gl_lead_rank eigenpairs only (kernels_gevd_lead.hip); reset by the call
    int gl_lead_done;        // set by apv_gevd_large: 1 = only the leading columns of U / entries of lambda were written
    void* lead_ws;           // workspace of apv_gevd_lead, owned
    void* comm;       // ncclComm_t
    int comm_rank, comm_world;
    hipStream_t comm_stream;      // the all-gather runs here so that it overlaps the next block's kernels
    hipEvent_t ev_ready;          // compute -> comm: the shard is written
    struct { const void* ptr; hipEvent_t ev; } gather_done[4];   // comm -> compute: shard buffer may be rewritten
    int gather_next;              // slot recycled next when all four track live buffers
    hipEvent_t ev_ag0, ev_ag1;    // timing events around the latest all-gather (comm stream)
    size_t ag_bytes;              // bytes this rank contributed to it
    int32_t* d_bar;               // one device word for apv_comm_barrier
    // update lanes (apv_set_update_streams): consecutive apv_update_dev launches alternate between two streams -- `stream` itself
    // (lane 0) and one more -- so that the tail of one launch (waves of its last round finishing one by one) runs beside the head
    // of the next.  `stream` is also the control stream: copies, timers and the gather's hand-over are ordered against the lanes by events.
    struct UpdateLane {
        hipStream_t s;
        hipEvent_t ev;            // recorded behind the lane's latest launch
        hipEvent_t ev_prev;       // ... and behind the one before it (the two handles swap at every launch)
        bool used;                // ev has been recorded at least once
        bool used_prev;           // ev_prev too
        bool need_fork;           // the control stream has had work since this lane last looked: wait for ev_fork first
        const void* rd[3];        // operand ranges of the latest launch: inputs ...
        size_t rd_bytes[3];
        const void* wr[3];        // ... and outputs
        size_t wr_bytes[3];
    } lane[2];
    void* d_Lspill_lane1;         // lane 1's own copy of the per-bin scratch slots (orders 33..64), allocated when pipelining is switched on
    int n_lanes;                  // 1: every launch on `stream` (the default), 2: pipelined
    int lane_next;
    bool ctrl_dirty;              // work has been put on the control stream that the lanes have not been ordered behind yet
    hipEvent_t ev_fork;
    std::string err;
};

void apv_stream_free(apv_handle* h);      // stream.hip
void apv_bb_free(apv_handle* h);          // stream_bb.hip
long apv_bb_not_converged(const apv_handle* h);   // stream_bb.hip: hops of the broadband stream that hit the sweep cap
void apv_gevd_large_free(apv_handle* h);  // kernels_gevd_large.hip
void apv_gevd_lead_free(apv_handle* h);   // kernels_gevd_lead.hip
int apv_fail(apv_handle* h, int code, const std::string& msg);
GevdParams apv_base_params(const apv_handle* h);

// kernels_gevd.hip
hipError_t apv_launch_gevd(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s, std::string* why);
int apv_gevd_reads_groups(const GevdParams& p, int compute_dtype, bool x_c128);      // bins per group of the spectra layout that launch reads (1: bin-major)
// HBM scratch the kernel that WILL run needs for K bins of order n (0 for most configurations): the order-64 kernel's slots when
// it is eligible, the float64 LDS kernel's parked Cholesky factor at orders 33..64 otherwise.  zones: 1 or 2 zone programs.
size_t apv_gevd_spill_bytes(int n, int K, int compute_dtype, int reg_mode, double reg_bright, double sweep_tol2, int zones);
bool apv_gevd64_eligible(int n, int reg_mode, double reg_bright, double sweep_tol2);

// kernels_gevd16m.hip: order-16 fast path (MFMA correlation / whitening / back-transform + register-resident
// Jacobi); hipErrorNotSupported when the problem does not qualify
hipError_t apv_launch_gevd16m(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s);
bool apv_gevd16m_takes_hops(const GevdParams& p, int compute_dtype, bool fused);      // would that launch accept p.n_hops > 1?

// kernels_gevd64.hip: order-64 float64 path (float32 block Jacobi on the f32 MFMA + float64 refinement on the f64 MFMA);
// hipErrorNotSupported when the problem does not qualify.  Needs p.Lspill with apv_gevd_spill_bytes() bytes.
hipError_t apv_launch_gevd64(const GevdParams& p, int compute_dtype, bool fused, hipStream_t s);
size_t apv_gevd64_slot_bytes();          // scratch per (zone program, bin)

// kernels_gevd_large.hip: real symmetric pairs of broadband order, f64, device pointers (see the file header)
int apv_gevd_large(apv_handle* h, int n, int batch, const double* d_A, const double* d_B, double reg,
                   const double* d_reg_scale, double* d_U, double* d_lam, const double* d_r, double mu, int V, const int* d_ranks, double* d_w, int32_t* h_status);

// kernels_gevd_lead.hip: the leading b eigenpairs of whitened matrices by Chebyshev-filtered subspace iteration (see the file header)
int apv_gevd_lead_block(int n, int rank);
// blocks of page-locked host memory handed out by apv_host_alloc (capi.hip)
void apv_host_blocks_note(const void* p, size_t bytes, bool add);
bool apv_host_block_contains(const void* p, size_t bytes);
int apv_gevd_lead(apv_handle* h, int n, int ne, int batch, int b, int rank, const double* C, const double* WT, double* d_U,
                  double* d_lam, const int* h_pd_flags, int* done);

// stream_bb.hip: d_out[i] = ||mats[i]||_2 (largest eigenvalue of a symmetric PSD n x n matrix, Lanczos), i < count <= 4
hipError_t apv_launch_norm2(int n, int count, const double* const* d_mats, double* d_out, hipStream_t s);

// kernels_corr.hip
hipError_t apv_launch_corr(int compute_dtype, int K, int M, int L, const float2* XB, const float2* XD,
                           const float2* d, void* RB, void* RD, void* r, hipStream_t s);

hipError_t apv_launch_corr_c128(int K, int M, int L, const double2* XB, const double2* XD, const double2* d, double2* RB,
                                double2* RD, double2* r, hipStream_t s);

// bf16 inputs ((re, im) bf16 pairs, 4 B per element), f32 accumulation on v_mfma_f32_32x32x16_bf16; L in {32, 64}
hipError_t apv_launch_corr_bf16(int K, int M, int L, const uint32_t* XB, const uint32_t* XD, const uint32_t* d,
                                float2* RB, float2* RD, float2* r, hipStream_t s);
hipError_t apv_launch_to_bf16(size_t count, const float2* in, uint32_t* out, hipStream_t s);

// kernels_stft.hip
hipError_t apv_launch_stft_analysis(int N, int n_ch, const float* x, float2* spec, hipStream_t s, std::string* why);
hipError_t apv_launch_istft_ola(int N, int H, int n_ch, const float2* spec, float* overlap, float* out,
                                hipStream_t s, std::string* why);
hipError_t apv_launch_stft_analysis_jobs(int f64, int N, int n_jobs, const void* const* x, const int* n_ch, void* const* spec,
                                         const long* stride_c, const long* stride_k, int ring_off, hipStream_t s,
                                         std::string* why);
hipError_t apv_launch_stft_analysis_strided(int N, int n_ch, const float* x, int ring_off, float2* spec,
                                            long stride_c, long stride_k, hipStream_t s, std::string* why);
hipError_t apv_launch_istft_ola_strided(int N, int H, int n_ch, const float2* spec, long stride_c, long stride_k,
                                        float* overlap, float* out, hipStream_t s, std::string* why);

bool apv_stft_size_ok(int N, std::string* why);
hipError_t apv_stft_prepare(int N, int f64);
// general forms: f64 = 0/1 selects float/double data; in_len < N zero-pads; use_win = 0 skips the sine window
hipError_t apv_launch_analysis(int f64, int N, int n_ch, const void* x, long x_stride, int in_len, int ring_off,
                               int use_win, void* spec, long stride_c, long stride_k, hipStream_t s, std::string* why);
// out_group > 0 (and dividing n_ch): the emitted hop is written sample-major in groups of out_group channels, [n_ch / out_group][H]
// [out_group], instead of channel-major [n_ch][H]; out_gstride > 0: elements between two groups (default H out_group)
hipError_t apv_launch_synthesis(int f64, int N, int H, int n_ch, const void* spec, long stride_c, long stride_k,
                                void* overlap, void* out, hipStream_t s, std::string* why, int out_group = 0, long out_gstride = 0);
// K1 (RIR convolution of a hop) by fast convolution, float (f64 = 0) or double data: see fir_fft_kernel
int apv_fir_fft_size(int f64, int P, int H);       // segment length F, 0 = use the direct form
hipError_t apv_launch_fir_spectra(int f64, int F, int n_ch, const void* x, int P, void* Hf, hipStream_t s, std::string* why);
hipError_t apv_launch_fir_input_spectra(int f64, int F, const void* x0, const void* x1, int in_len, void* Xf, hipStream_t s);
hipError_t apv_launch_fir_chunk_spectra(int f64, int F, int P, int H, int n_hops, const void* hist0, const void* hist1,
                                        const void* pin, void* Xf, hipStream_t s);
// upd != nullptr: the launch also carries the hop's input update (what apv_launch_input_update does), see FirFftJobs
struct ApvInputUpdate {
    const void* old_hist[2];
    void* new_hist[2];
    const void* xin;      // pinned host [2][H]
    void* inblk;          // [2][N] rings
    int pad;
};
// n_part > 1: uniformly partitioned (partitions of H taps, F = 2 H): Hf[j] is [n_part][C_j][F/2 + 1], Xf[j] points at the job's
// signal inside [n_part][2][F/2 + 1]
hipError_t apv_launch_fir_fft_jobs(int f64, int F, int n_jobs, const void* const* Hf, const void* const* Xf, void* const* resp,
                                   const int* n_ch, int P, int H, int N, int ring_off, const ApvInputUpdate* upd, hipStream_t s,
                                   int n_part = 1);
hipError_t apv_launch_fir_input_spectra_parts(int f64, int F, int n_part, const void* x0, const void* x1, void* Xf, hipStream_t s);
hipError_t apv_launch_fir_spectra_part(int f64, int F, int n_ch, const void* x, long x_stride, int taps, void* Hf, hipStream_t s,
                                       std::string* why);
// uniformly partitioned K1 for (P, H): number of partitions (segments of 2 H samples), 0 when it does not apply
int apv_fir_partitions(int f64, int P, int H);

// whole-signal path, a chunk of hops per launch (kernels_stft.hip / kernels_stream.hip; see process_signal_chunked_t in stream.hip)
hipError_t apv_launch_stft_analysis_chunk(int f64, int N, int n_jobs, const void* const* x, const int* n_ch, void* const* spec,
                                          const long* stride_c, const long* stride_k, long x_stride, long x_hop, const long* spec_hop,
                                          int n_hops, hipStream_t s, std::string* why);
hipError_t apv_launch_fir_fft_chunk(int f64, int F, int n_jobs, const void* const* Hf, const void* const* Xf, long x_hop_stride,
                                    void* const* resp, const int* n_ch, int P, int H, int row_stride, int pos0, int n_hops,
                                    hipStream_t s);
hipError_t apv_launch_rows_copy(int f64, int rows, int len, void* dst, long ds, int d0, int dmod, const void* src, long ss, int s0,
                                int smod, hipStream_t s);
hipError_t apv_launch_chunk_inputs(int f64, int P, int H, int N, int nc, int pad, int RL, const void* const old_hist[2],
                                   void* const new_hist[2], const void* pin, void* inL, hipStream_t s);

// kernels_stream.hip
// y = FIR(rir, x) for one hop, appended to the ring response buffers:
//   resp[c*N + ((N-H+n + ring_off) & (N-1))] = sum_p rir[p*C + c] * xhist[P-1 + n - p],  n < H, c < C
hipError_t apv_launch_fir_hop(int C, int P, int H, int N, int ring_off, const float* rir, const float* xhist,
                              float* resp, hipStream_t s);
struct FirJob { const float* rir; const float* xh; float* resp; int C; };
struct FirJobs { FirJob j[6]; int n; };
// all FIR jobs of a hop in one launch, on the matrix cores (v_mfma_f32_32x32x2_f32, implicit Toeplitz operand)
hipError_t apv_launch_fir_jobs(const FirJobs& jobs, int P, int H, int N, int ring_off, hipStream_t s);
// float64 jobs of one hop on v_mfma_f64_16x16x4_f64 (both stream modes); tile0 is filled by the launcher
constexpr int FIR_JOBS_D = 6;
struct FirJobsD {
    const double* rir[FIR_JOBS_D];     // [P][C_j]
    const double* xh[FIR_JOBS_D];      // input history of the job's signal
    double* resp[FIR_JOBS_D];          // [C_j][N] ring
    int C[FIR_JOBS_D];
    int tile0[FIR_JOBS_D + 1];         // first channel tile of each job
};
hipError_t apv_launch_fir_jobs_f64(FirJobsD jobs, int njobs, int P, int H, int N, int ring_off, hipStream_t s);
int apv_fir_pad_f64();
// f64 = 0: c64 spectra / float weights; 1: c128 spectra / double weights (bin-major [K][M])
hipError_t apv_launch_perceptual_weights(int f64, int K, int M, int nch, const void* spec, const double* G2, const double* G2T,
                                         double Cs, double Ca, double Leff, int N, int norm_mode, void* W, hipStream_t s);
hipError_t apv_launch_scale_spectra(int f64, int K, int C, int L, void* spec, const void* W, hipStream_t s);
hipError_t apv_launch_ungroup_spectra(int f64, int K, int C, int g, const void* in, void* out, hipStream_t s);
hipError_t apv_launch_perceptual_weights_f64(int K, int M, int nch, const double2* spec, const double* G2,
                                             const double* G2T, double Cs, double Ca, double Leff, int N, int norm_mode,
                                             double* W, hipStream_t s);
hipError_t apv_launch_scale_spectra_cm_f64(int K, int C, int L, double2* spec, const double* W, hipStream_t s);
hipError_t apv_launch_input_update(int f64, int P, int H, int pad, int N, int ring_off, const void* const old_hist[2],
                                   void* const new_hist[2], const void* xin, void* inblk, hipStream_t s);
int apv_fir_pad();
// out[ch][k] = in_spec[k] * filt(ch, k): ch < n_filt channels taken from the bin-major filter bank
// w[k][n_filt] (c64 or c128), remaining channels from the channel-major table tgt[ch - n_filt][k]
hipError_t apv_launch_apply_jobs(int K, int n_jobs, const void* const* in_spec, const void* const* w, const void* const* tgt,
                                 void* const* out, const int* n_filt, const int* n_tgt, int w_c128, int spec_f64, hipStream_t s);
