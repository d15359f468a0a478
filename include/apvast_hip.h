/*
 * apvast_hip.h -- C ABI of the MI355X-native AP-VAST filter engine.
 *
 * This is the drop-in boundary for the per-block, per-subband hot path of the
 * reference (macoustics/ap-vast-unofficial).  The reference has no native code
 * and therefore no FFI of its own; each entry point below replaces a span of
 * NumPy/SciPy calls in Python/apvast.py, cited as "replaces:".  The host side
 * that binds it is ap_vast_unofficial_amd/_capi.py (ctypes); INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns an int status: 0 = OK,
 *     negative = error (apv_last_error() gives the text).
 *   - complex numbers are interleaved (re, im) pairs: `float[2]` for c64,
 *     `double[2]` for c128.
 *   - a handle owns one HIP device, one stream, its workspaces and (optionally)
 *     one RCCL communicator.  Handles are thread-compatible, not thread-safe.
 *   - pointer arguments named h_* are HOST pointers, d_* are DEVICE pointers
 *     obtained from apv_dev_alloc() (or any hipMalloc'd memory on the handle's
 *     device).  No entry point allocates on the per-block path: apv_update_dev, apv_process_block*, apv_process_signal*
 *     and apv_bb_process_block use buffers sized by apv_create / apv_*_init (apv_process_signal sets up its second
 *     spectra set and pinned staging at its first call); the on-demand readers (apv_stream_get_statistics) allocate their
 *     work space once and keep it; the host-buffer conveniences (apv_update, apv_jdiag_*, apv_predict_pressure,
 *     apv_vast_static) stage through device memory of their own.
 *
 * Data layout (HBM), subband mode -- bin-major so that one bin's control-point
 * matrix is one contiguous, coalesced slab:
 *   X_B, X_D : [K][M][L] c64   bright / dark control-point spectra of the K bins
 *                              owned by this handle; X[k] = spectra[k].T of
 *                              apvast.py:239-262, rows = control points (mics),
 *                              columns = loudspeakers (fastest).
 *   d        : [K][M]    c64   target spectrum at the bright control points
 *                              (apvast.py:199-209).
 *   w        : [K][nV][L] c64 (or c128)  VAST filter per bin and per rank V.
 *   lam      : [K][L]    f32 (or f64)    generalized eigenvalues, descending.
 *   status   : [K]       i32   0 OK; 1 = loaded R_D not positive definite
 *                              (numpy.linalg.LinAlgError at apvast.py:24);
 *                              2 = eigen-iteration did not converge.
 */
#ifndef APVAST_HIP_H
#define APVAST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APV_ABI_VERSION 2
#define APV_MAX_RANKS 64     /* number of simultaneously produced ranks V (nV) */
#define APV_MAX_N 64         /* largest GEVD order n = L handled on-chip */

/* status codes */
#define APV_OK 0
#define APV_ERR_ARG (-1)        /* invalid argument / unsupported size */
#define APV_ERR_HIP (-2)        /* HIP runtime error */
#define APV_ERR_NOT_PD (-3)     /* at least one bin: R_D + reg not positive definite */
#define APV_ERR_NO_CONVERGE (-4)/* at least one bin: eigen-iteration hit the sweep cap */
#define APV_ERR_RCCL (-5)       /* RCCL error / communicator not initialised */
#define APV_ERR_STATE (-6)      /* unknown state name / wrong size */

/* arithmetic the GEVD + filter run in (inputs are always c64) */
#define APV_F32 0
#define APV_F64 1

/* dark-matrix loading, apvast.py:22-27 */
#define APV_REG_ABS 0           /* B + reg_dark * I              (EXPERIMENTAL_REGULARIZATION=True) */
#define APV_REG_REL 1           /* B + reg_dark * ||B||_2 * I    (else-branch; MATLAB diagonalLoading) */

typedef struct apv_handle apv_handle;

#define APV_DIALECT_PYTHON 0   /* apvast.py: skipped sample in the data matrix, no normalisation, absolute dark loading, ranks 1..V */
#define APV_DIALECT_MATLAB 1   /* apVast.m: contiguous Hankel matrix, R and r / ((S-J+1) M), loading relative to ||R||_2, rank list */

typedef struct apv_config {
    int32_t abi_version;      /* APV_ABI_VERSION */
    int32_t device;           /* HIP device ordinal */
    int32_t n_bins;           /* K  : bins owned by this handle (its shard) */
    int32_t n_srcs;           /* L  : loudspeakers = GEVD order n (<= APV_MAX_N) */
    int32_t n_mics;           /* M  : control points per zone */
    int32_t n_ranks;          /* nV : how many ranks V are produced (<= APV_MAX_RANKS; the reference emits every rank 1..V, apvast.py:406-422) */
    int32_t ranks[APV_MAX_RANKS]; /* the V list, each 1..L, ascending */
    int32_t compute_dtype;    /* APV_F32 | APV_F64 */
    int32_t out_c128;         /* 0: w c64 / lam f32;  1: w c128 / lam f64 */
    int32_t reg_mode;         /* APV_REG_ABS | APV_REG_REL */
    double  reg_dark;         /* 1e-7 in the Python dialect (apvast.py:23) */
    double  reg_bright;       /* relative bright loading (apVast.m:552-569); 0 in the Python dialect */
    double  mu;               /* trade-off parameter (apvast.py:49, 410) */
    int32_t max_sweeps;       /* Jacobi sweep cap; 0 = default.  Broadband mode: a value > 0 also selects the complete block-Jacobi solve for every hop (0: the per-hop path computes the leading eigenpairs the filters use, the rest when lambda / U are read) */
    int32_t block_size;       /* N : STFT length for the streaming entry points (0 = kernel-level use only) */
    int32_t hop_size;         /* H */
    int32_t n_zones;          /* streaming: bit mask of zone programs, 1 = A, 2 = B (run_A/run_B, apvast.py:53-54) */
    int32_t debug_stop;       /* profiling aid: stop the fused kernel after stage n (0 = run everything) */
    int32_t dialect;          /* APV_DIALECT_PYTHON | APV_DIALECT_MATLAB: the broadband stream's statistics/loading/rank conventions */
    int32_t frontend;         /* streaming (subband) front-end precision -- RIRs, FIR, rings, STFT, spectra, overlap-add:
                                 0 = follow compute_dtype (APV_F64: everything float64, as the reference's lfilter / rfft /
                                 irfft are, apvast.py:171-192, 202-203, 461-496), 1 = float32, 2 = float64 */
    int32_t out_layout;       /* streaming (subband) outputs: 0 = channel-major h_out [n_out][H]; 1 = sample-major in groups of L channels
                                 (one zone program and rank each), h_out [n_out / L][H][L] -- the (hop, loudspeaker) arrays the
                                 reference's caller receives (apvast.py:498-504), so the host binding reshapes instead of
                                 transposing; apv_process_signal* then writes [n_out / L][n_hops * H][L] */
    int32_t reserved[4];
    double  sweep_tol2;       /* Jacobi stop threshold: a sweep whose pivots satisfy sum |c_pq|^2 <= sweep_tol2 ||C||_F^2 is the
                                 last one (quadratic convergence leaves ~sweep_tol2^2 behind).  0 = default (1e-16 in float64:
                                 csrc/gevd16_common.h Prec<double>, csrc/kernels_gevd.hip Tol<double>; 1e-8 in float32);
                                 apv_jdiag_* always iterate to 1e-17 */
} apv_config;

/* ---- lifetime ---------------------------------------------------------- */
int  apv_create(const apv_config* cfg, apv_handle** out);     /* replaces: apvast.__init__ allocation, apvast.py:115-151 */
int  apv_destroy(apv_handle* h);
const char* apv_last_error(const apv_handle* h);               /* h may be NULL: last create() error */
int  apv_abi_version(void);

/* ---- device memory / stream plumbing ----------------------------------- */
int  apv_dev_alloc(apv_handle* h, size_t bytes, void** d_ptr);
int  apv_dev_free(apv_handle* h, void* d_ptr);
int  apv_memcpy_h2d(apv_handle* h, void* d_dst, const void* h_src, size_t bytes);   /* async on the handle's stream */
int  apv_memcpy_d2h(apv_handle* h, void* h_dst, const void* d_src, size_t bytes);   /* async on the handle's stream */
int  apv_sync(apv_handle* h);                                  /* hipStreamSynchronize */
/* HIP-event timer on the handle's stream (bench.py's roofline leg) */
int  apv_timer_start(apv_handle* h);
int  apv_timer_stop(apv_handle* h, float* elapsed_ms);         /* synchronises on the stop event */

/* ---- the hot path, device-resident ------------------------------------- */
/* One block of subband filter updates: for each owned bin k
 *   R_B = X_B^H X_B, R_D = X_D^H X_D, r = X_B^H d          replaces apvast.py:329-364 (update_statistics)
 *   U, lam = jdiag(R_B, R_D)                                replaces apvast.py:20-36, 378-387
 *   w_V = sum_{i<V} (u_i^H r)/(lam_i+mu) u_i                replaces apvast.py:406-414
 * R, U never touch HBM.  d_lam / d_status may be NULL.
 */
int  apv_update_dev(apv_handle* h, const void* d_XB, const void* d_XD, const void* d_d,
                    void* d_w, void* d_lam, int32_t* d_status);

/* Pipelining of consecutive apv_update_dev calls (the caller of apvast.py:153-165 runs one block after the other; blocks of
 * different streams or hops are independent).  n = 2: launches alternate between two streams of the handle's, so that the last
 * waves of one launch finish beside the first waves of the next (+6 % updates/s at BASELINE config 2).  Results are those of
 * n = 1 bit for bit.  Ordering stays the library's business: a launch whose operands overlap the other lane's launch in flight
 * (same output buffer, an input that the other writes) waits for it; copies, timers, apv_sync, apv_update and the all-gather see
 * every launch queued before them; launches queued after a copy see the copy.  Orders 33..64 park per-bin state in scratch slots:
 * the second stream gets slots of its own (allocated by this call: the per-block path still allocates nothing); the split float32
 * update (orders 32 / 64 in APV_F32) shares one scratch matrix and keeps to one stream.  n = 1 (the default): every launch on the
 * handle's stream. */
int  apv_set_update_streams(apv_handle* h, int32_t n);

/* Same from caller-owned host buffers (copies in, runs, copies out, syncs). */
int  apv_update(apv_handle* h, const float* h_XB, const float* h_XD, const float* h_d,
                void* h_w, void* h_lam, int32_t* h_status);

/* K5': correlation only -> R_B, R_D [K][L][L], r [K][L] in the compute dtype
 * (c64 for APV_F32, c128 for APV_F64), device-resident.   replaces apvast.py:329-364 */
int  apv_corr_dev(apv_handle* h, const void* d_XB, const void* d_XD, const void* d_d,
                  void* d_RB, void* d_RD, void* d_r);

/* The same contraction from bf16 inputs ((re, im) bf16 pairs, 4 bytes per element, same [K][M][L] layout), f32
 * accumulation on the bf16 matrix cores; R_B, R_D [K][L][L] and r [K][L] come out as c64.  n_srcs in {32, 64}.
 * (BASELINE config 5: fp32 vs bf16 correlation accumulation.)  apv_to_bf16_dev converts `count` c64 elements. */
int  apv_corr_bf16_dev(apv_handle* h, const void* d_XB, const void* d_XD, const void* d_d,
                       void* d_RB, void* d_RD, void* d_r);
int  apv_to_bf16_dev(apv_handle* h, size_t count, const void* d_c64, void* d_bf16);

/* K6-K10: GEVD + filter from explicit R_B, R_D, r (compute dtype).   replaces apvast.py:378-414 */
int  apv_gevd_vast_dev(apv_handle* h, const void* d_RB, const void* d_RD, const void* d_r,
                       void* d_w, void* d_lam, int32_t* d_status);

/* Module-level jdiag drop-in (apvast.py:20-36), batched: `batch` pairs of n x n Hermitian
 * (A, B) in c128 (row-major), B loaded with reg per the handle's reg_mode/reg_dark.
 * U: [batch][n][n] c128 (columns = eigenvectors, descending), lam: [batch][n] f64.
 * Host buffers. n <= APV_MAX_N. */
int  apv_jdiag_batched(apv_handle* h, int32_t n, int32_t batch, const double* h_A, const double* h_B,
                       double* h_U, double* h_lam, int32_t* h_status);

/* The same for REAL symmetric pairs of broadband order (n = filter_length x loudspeakers: 256 at cfg1, 800 with
 * the parameters of make_python_test.m), n <= 2048: A, B, U are [batch][n][n] f64 row-major, lam [batch][n].
 * Matrices live in HBM; one launch per elimination step and per Jacobi round.   replaces apvast.py:20-36 at
 * the sizes of its call sites apvast.py:380, 382 */
int  apv_jdiag_large(apv_handle* h, int32_t n, int32_t batch, const double* h_A, const double* h_B,
                     double* h_U, double* h_lam, int32_t* h_status);
/* The leading `rank` eigenpairs of real symmetric pairs of broadband order -- what the filters consume: apvast.py:406-414 uses
 * U[:, :V] and diag(D)[:V] of the jdiag call at apvast.py:380, 382.  U: [batch][n][rank] (columns = eigenvectors, descending,
 * U^T (B + reg I) U = I), lam: [batch][rank].  Chebyshev-filtered subspace iteration on the whitened matrix
 * (kernels_gevd_lead.hip); h_info[z] = 0: that solver converged, 1: the call fell back to the complete block-Jacobi solve of
 * apv_jdiag_large (flat spectrum beyond the cut, rank too large for a block of 64, Gram breakdown) -- the results are
 * valid either way.  h_info may be NULL.                      replaces apvast.py:20-36 + the slices of 406-414 */
int  apv_jdiag_leading(apv_handle* h, int32_t n, int32_t batch, int32_t rank, const double* h_A, const double* h_B,
                       double* h_U, double* h_lam, int32_t* h_info);
/* Complex Hermitian pairs beyond the per-bin orders, n <= 1024 (apvast.py:20-36 takes any order; apv_jdiag_batched stops at
 * 64): A, B, U are [batch][n][n] complex128 row-major, lam [batch][n] f64.  Runs the real symmetric solver above on the
 * embedding [Re -Im; Im Re] of order 2n and keeps n of its 2n eigenvectors that are independent over C
 * (kernels_jdiag_cplx.hip).  status 3 / APV_ERR_NO_CONVERGE: an eigenvalue cluster of ~40 or more coincident values.
 *                                                         replaces apvast.py:20-36 */
int  apv_jdiag_large_c128(apv_handle* h, int32_t n, int32_t batch, const void* h_A, const void* h_B,
                          void* h_U, double* h_lam, int32_t* h_status);

/* ---- STFT stages (K2-K4) ------------------------------------------------ */
/* spectra[c][k] = rfft(window * x[c][:])  for `n_ch` channels of length N (f32 in, c64 out,
 * both channel-major, device).      replaces apvast.py:202-203, 246-255, 430-431 */
int  apv_stft_analysis_dev(apv_handle* h, int32_t n_ch, const float* d_x, void* d_spec);
/* overlap[c] = shift(overlap[c], H) + window * irfft(spec[c]); out[c][0:H] = overlap[c][0:H].
 *                                   replaces apvast.py:212-225, 265-293, 457-504 */
int  apv_istft_ola_dev(apv_handle* h, int32_t n_ch, const void* d_spec, float* d_overlap, float* d_out);

/* ---- streaming composition (one call = one hop) -------------------------- */
/* Upload the room impulse responses and allocate all streaming state.  h_rir_A / h_rir_B: (rir_len, L, M)
 * float64, C order -- exactly what scipy.io.loadmat('rirs.mat') gives after np.ascontiguousarray
 * (make_python_test.m:4,18).  The handle must have been created with block_size, hop_size,
 * n_bins = block_size/2+1 and n_zones = bit mask (1 = zone A, 2 = zone B: run_A/run_B, apvast.py:53-54).
 *                                                         replaces apvast.__init__, apvast.py:97-151 */
int  apv_stream_init(apv_handle* h, int32_t rir_len, const double* h_rir_A, const double* h_rir_B,
                     int32_t reference_index_A, int32_t reference_index_B, int32_t modeling_delay);
/* One hop of both input signals (H float32 samples each).  h_out: [n_out][H] float32 with channels
 * [zone A: nV x L][zone B: nV x L] (zones that run) followed by [A_t: L][B_t: L]  (cfg.out_layout = 1: the same groups of L
 * channels, each written [H][L]).
 *                                                         replaces process_input_buffers, apvast.py:153-165 */
int  apv_process_block(apv_handle* h, const float* h_in_A, const float* h_in_B, float* h_out);
/* The same with float64 samples on the host side (lossless with the float64 front-end; either entry point works
 * with either front-end, converting on the host).  A hop in which some bin reached the Jacobi sweep cap returns
 * APV_ERR_NO_CONVERGE after h_out has been written.      replaces process_input_buffers, apvast.py:153-165 */
int  apv_process_block_f64(apv_handle* h, const double* h_in_A, const double* h_in_B, double* h_out);
/* n_hops consecutive hops in one call: h_in_A / h_in_B hold n_hops * H samples, h_out is [n_hops][n_out][H] with the
 * channel order of apv_process_block.  Sample for sample the result is that of n_hops calls of apv_process_block (same
 * kernels, same operands, same order within a hop); the hops are pipelined: RIR convolution + analysis of hop h+1 run on a second
 * stream beside the joint diagonalisation of hop h, synthesis and copy back on a third, and the host stages / converts
 * one chunk of 16 hops while the device runs the next.  A hop with a bin that is not positive definite returns APV_ERR_NOT_PD once the
 * chunks in flight (its own and at most one more) have run (the message names the hop; the stream's state is then that
 * after the last hop run); APV_ERR_NO_CONVERGE is returned after all hops, every output written.
 *                        replaces the hop loop around processInputBuffer, main.m:52-62 / make_python_test.m:44-51 */
int  apv_process_signal(apv_handle* h, int32_t n_hops, const float* h_in_A, const float* h_in_B, float* h_out);
int  apv_process_signal_f64(apv_handle* h, int32_t n_hops, const double* h_in_A, const double* h_in_B, double* h_out);
/* 1 if the stream's front-end (and therefore its named state arrays) is float64, 0 if float32, -1 without a stream */
int  apv_stream_is_f64(apv_handle* h);
/* Per-bin statistics and eigenvectors of the CURRENT hop of a streaming handle, recomputed in float64 on demand from
 * the hop's control-point spectra (the per-hop path never writes R or U to HBM).  zone 0 = A (bright A->A, dark A->B),
 * 1 = B.  h_RB, h_RD, h_U [K][L][L] c128 (U: columns = eigenvectors, descending), h_r [K][L] c128, h_lam [K][L] f64; any
 * of them may be NULL.       replaces the attributes R_*, r_*, U_*, lambda_* of apvast.py:368-387, per bin */
int  apv_stream_get_statistics(apv_handle* h, int32_t zone, double* h_RB, double* h_RD, double* h_r, double* h_U,
                               double* h_lam);
/* hops so far (either stream mode) in which some bin reached the Jacobi sweep cap; each such hop also returned
 * APV_ERR_NO_CONVERGE with its outputs written and the stream advanced: a warning to the caller, not a lost hop */
long apv_stream_not_converged(apv_handle* h);
/* Perceptual ("AP") weighting of the control-point and target spectra, evaluated per block on the device from the
 * target spectra (van de Par 2005 model as carried by the reference's MATLAB twin).  h_G2 [K][n_channels] float64
 * = squared outer/middle-ear x gammatone responses; n_channels = 0 switches it off (all-ones, apvast.py:326-327).
 *                         replaces update_perceptual_weighting, apvast.py:313-324 / apVast.m:386-408 */
int  apv_stream_set_perceptual(apv_handle* h, int32_t n_channels, const double* h_G2, double Cs, double Ca,
                               double Leff, int32_t normalisation);
/* Named state arrays for fixtures / checkpoint-resume (names: see stream.hip), in the front-end precision:
 * float32 / complex64, or float64 / complex128 with the float64 front-end.          apvast.py:115-151 */
int  apv_state_bytes(apv_handle* h, const char* name, size_t* bytes);
int  apv_get_state(apv_handle* h, const char* name, void* h_dst, size_t bytes);
int  apv_set_state(apv_handle* h, const char* name, const void* h_src, size_t bytes);

/* ---- broadband (time-domain) mode: the reference's own algorithm, float64 ---------------------------- */
/* Rank list of the next apv_bb_init: apVast.m:527-549 takes a vector of ranks and emits one solution per entry
 * (ascending, each 1..J L); n_ranks = 0 restores "every rank 1..number_of_eigenvectors" (apvast.py:406-422).
 * With cfg.dialect = APV_DIALECT_MATLAB the stream also follows apVast.m:410-456 (contiguous data matrix, R and r
 * divided by (S-J+1) M), apVast.m:597-602 (one target reference per zone) and, with cfg.reg_mode = APV_REG_REL,
 * apVast.m:552-569 (bright += reg_bright ||R||_2, dark += reg_dark ||R||_2 in place; spectral norm = largest Ritz
 * value of 96 Lanczos steps). */
int  apv_bb_set_rank_list(apv_handle* h, int32_t n_ranks, const int32_t* ranks);
/* One real (J L) x (J L) pair per zone per hop from `statistics_buffer_length` samples, J-tap filters, every rank
 * 1..V (apvast.py:329-422) or the registered rank list.  The handle needs block_size, hop_size, n_srcs, n_mics,
 * n_zones, mu, dialect, reg_mode and reg_dark (reg_mode = APV_REG_ABS: dark + reg_dark I inside the joint
 * diagonalisation, apvast.py:22-24; APV_REG_REL: see above); n_bins / ranks are not used.  J L <= 2048,
 * block_size <= 4096.  A failed factorisation (APV_ERR_NOT_PD) leaves the input/response/statistics buffers
 * advanced by the hop and the output overlap buffers untouched.
 *                                                         replaces apvast.__init__, apvast.py:40-151 */
int  apv_bb_init(apv_handle* h, int32_t rir_len, const double* h_rir_A, const double* h_rir_B,
                 int32_t reference_index_A, int32_t reference_index_B, int32_t modeling_delay,
                 int32_t filter_length, int32_t statistics_buffer_length, int32_t number_of_eigenvectors);
/* One hop (H float64 samples per signal).  h_out: [n_out][H] float64, channels as for apv_process_block with
 * nV = the number of ranks kept; with cfg.out_layout = 1: [n_out / L][H][L], one (hop, loudspeaker) array per zone program
 * and rank as process_input_buffers returns them (apvast.py:498-504).   replaces process_input_buffers, apvast.py:153-165 */
int  apv_bb_process_block(apv_handle* h, const double* h_in_A, const double* h_in_B, double* h_out);
/* n_hops consecutive hops in one call: h_in_A / h_in_B hold n_hops * H samples, h_out is [n_hops][n_out][H] with the channel
 * order of apv_bb_process_block -- with cfg.out_layout = 1: [n_out / L][n_hops * H][L], every group's whole signal as one
 * (sample, loudspeaker) array.  Each group of hops reaches the host in one copy: by DMA into a page-locked h_out
 * (apv_host_alloc), else through a staging set that the host empties while the device works on the next group.  A hop's statistics never depend on an earlier hop's filters, so the joint diagonalisations
 * of up to 16 consecutive hops (8 at large orders) are solved as ONE batch (the dependent launches of a single n = 256 pair leave most of the chip
 * idle); rings, overlap buffers and outputs advance hop by hop as in the per-hop call.  Outputs equal those of n_hops calls
 * of apv_bb_process_block up to the rounding of the eigen-iteration (a batch sweeps until its slowest member has converged).
 *                        replaces the hop loop of main.m:52-62 / make_python_test.m:44-51 around apvast.py:153-165 */
int  apv_bb_process_signal(apv_handle* h, int32_t n_hops, const double* h_in_A, const double* h_in_B, double* h_out);
/* Page-locked host memory for result arrays (hipHostMalloc, visible to every device): device-to-host copies into it are
 * asynchronous DMA transfers.  No reference counterpart (numpy allocates the reference's results, apvast.py:433-443). */
int  apv_host_alloc(void** p, size_t bytes);
int  apv_host_free(void* p);
/* perceptual weighting for the broadband stream; arguments as for apv_stream_set_perceptual */
int  apv_bb_set_perceptual(apv_handle* h, int32_t n_channels, const double* h_G2, double Cs, double Ca, double Leff,
                           int32_t normalisation);
/* float64 state arrays by name (stream_bb.hip); `count` = number of doubles */
int  apv_bb_get_state(apv_handle* h, const char* name, double* h_dst, size_t count);
int  apv_bb_set_state(apv_handle* h, const char* name, const double* h_src, size_t count);

/* ---- evaluation and the static solver, float64 ------------------------- */
/* p[t][m] = sum_s filter(rir[:, s, m], 1, x[:, s])[t].  h_x [T][L], h_rir [P][L][M], h_out [T][M].
 *                                                         replaces Matlab/ControlMethods/predictPressure.m:1-17 */
int  apv_predict_pressure(apv_handle* h, int32_t T, int32_t L, int32_t M, int32_t P, const double* h_x,
                          const double* h_rir, double* h_out);
/* Static (signal-independent) VAST filters from the impulse responses: h_gB [Nb][P][L], h_gD [Nd][P][L]
 * (vast.m's (mics, rirLength, sources) layout), h_w [J*L] (speaker-major taps).
 *                                                         replaces Matlab/ControlMethods/vast.m:1-97 */
int  apv_vast_static(apv_handle* h, int32_t Nb, int32_t Nd, int32_t P, int32_t L, int32_t J, int32_t modeling_delay,
                     int32_t reference_index, int32_t V, double mu, const double* h_gB, const double* h_gD, double* h_w);

/* ---- multi-GPU: bins sharded across ranks, one RCCL all-gather ---------- */
int  apv_comm_unique_id(char id_out[128]);                                /* rank 0 calls, then broadcasts */
int  apv_comm_init(apv_handle* h, const char id[128], int32_t rank, int32_t world);
/* gathers every rank's w shard [K][nV][L] into d_w_all [world*K][nV][L] (device) over xGMI.  Asynchronous:
 * it waits for the kernels already queued on the handle's stream, then runs on a second stream so that the
 * next apv_update_dev (into a different shard buffer) overlaps it; an update into the SAME shard buffer waits
 * for the gather.  apv_sync() waits for both streams. */
int  apv_allgather_filters_dev(apv_handle* h, const void* d_w_shard, void* d_w_all);
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank); apv_comm_init already fails when they differ from
 * what it was given.  user_rank may be NULL. */
int  apv_comm_count(apv_handle* h, int32_t* n_ranks, int32_t* user_rank);
/* device time of the latest all-gather (HIP events on the communication stream) and the bytes this rank sent */
int  apv_comm_last_gather(apv_handle* h, float* elapsed_ms, size_t* bytes_per_rank);
/* all ranks of the communicator meet here (one-word ncclAllReduce); also drains the handle's compute stream */
int  apv_comm_barrier(apv_handle* h);
/* Diagnostics (tools/probes/stage_stamps.py; never set in normal use): with a device buffer of [zones][K][16] uint64 registered,
 * the order-16 update runs its diagnostic instantiation, in which lane 0 of every bin's wave stores the shader clock (s_memtime)
 * at the stage boundaries; NULL switches it off again.  The stamps go to this buffer only; no result depends on them. */
int  apv_debug_set_stamps(apv_handle* h, void* d_stamps);
/* hipDeviceSynchronize on the handle's device, and what that device is */
int  apv_device_sync(apv_handle* h);
int  apv_device_info(apv_handle* h, char name_out[128], int32_t* n_cus, int32_t* clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* APVAST_HIP_H */
