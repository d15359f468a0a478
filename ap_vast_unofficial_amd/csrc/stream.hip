// Streaming composition of the subband hot path: one call = one hop of both input signals.
//
//   apv_stream_init   : upload RIRs, allocate every buffer once      reference Python/apvast.py:97-151
//   apv_process_block : per hop                                       apvast.py:153-165
//       K1  RIR convolution of the hop to all control points          apvast.py:167-194
//       K2  sine-window analysis STFT of the response / target rings  apvast.py:197-203, 244-255
//       K5'-K10 per-bin correlate + jdiag + VAST filter, per zone     apvast.py:329-414 (subband form)
//       K3  output spectra = input spectrum x filter bank             apvast.py:445-452
//       K4  inverse STFT, window, overlap-add, emit first H samples   apvast.py:457-504
//
// Device layouts: control-point channel c = m*L + l (loudspeaker fastest), so a bin's control-point
// matrix X[k] = [M][L] is one contiguous slab of the bin-major spectra [K][M*L].
#include "apv_internal.h"

#include <algorithm>
#include <chrono>
#include <functional>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int CK_NB = 4;      // back streams of the chunked whole-signal path
constexpr int CK_NS = 3;      // pinned staging slots (chunks) of that path

struct apv_stream {
    int N, H, K, L, M, C, P, nV, zones, pad;
    int f64;                      // precision of the whole front-end: RIRs, rings, spectra, overlap buffers, outputs
    size_t esz;                   // bytes per real sample (4 | 8); a complex spectrum element is 2 esz
    int ring_off;                 // physical index of logical sample 0 in every ring
    int cur;                      // which input-history buffer is current
    int n_out;                    // synthesis channels: zones*nV*L filtered + 2*L target
    int out_group;                // cfg.out_layout = 1: L (hops are emitted [n_out / L][H][L]); 0 = channel-major [n_out][H]
    void* rir[2];                 // [P][C]  zone A, zone B
    void* trir[2];                // [P][M]  target RIRs (reference loudspeaker, delayed)
    void* xhist[2][2];            // [buf][signal][keep+H+pad]
    void* resp[4];                // [C][N] rings: A->A, A->B, B->A, B->B
    void* tresp[2];               // [M][N] rings: target A, target B
    void* inblk;                  // [2][N] rings: input blocks
    void* X[4];                   // control-point spectra: bin-major [K][C], or grouped [Kp / xg][C][xg] (see xg); Kp C elements each
    int xg, xg_default;           // bins per group of the spectra layout (1 = bin-major).  4 where the joint diagonalisation that will
                                  // run reads groups (float64 front-end, order 16, apv_gevd_reads_groups) and no perceptual weighting
                                  // is on (its kernels scale bin-major spectra): a transform then fills 64-byte lines
    int Kp;                       // K rounded up to a multiple of 8: bins a spectra set is allocated for
    void* tspec[2];               // [K][M]
    void* inspec;                 // [2][K]
    void* w[2];                   // [K][nV][L] per zone
    void* lam[2];                 // [K][L]
    int32_t* status[2];           // [K]
    void* tgt;                    // [L][K] target filter spectra (shared by A_t and B_t, apvast.py:389-390)
    void* outspec;                // [n_out][K]
    void* outov;                  // [n_out][N]
    void* out;                    // [n_out][H] samples, then the [2][K] status words of the hop: one copy back
    // perceptual weighting (off when nch == 0)
    int nch, norm_mode;
    double Cs, Ca, Leff;
    double* G2;                   // [K][nch]
    double* G2T;                  // [nch][K]
    void* Wgt[2];                 // [K][M] per zone
    // pinned staging + one captured hipGraph per phase of the (ring offset, history buffer) cycle
    void* pin_in;                 // [2][H]
    void* pin_out;                // [n_out][H] samples + [2][K] status words
    int period;                   // hops after which (ring_off, cur) repeat; 0 = graphs off
    // whole-signal path (apv_process_signal): a second set of the hop's spectra, a second stream and four events, so
    // that the front half (FIR + analysis) of hop h+1 runs beside the back half (GEVD + synthesis) of hop h
    void* X1[4];                  // like X, tspec, inspec: hops alternate between the two sets
    void* tspec1[2];
    void* inspec1;
    hipStream_t front;            // front-half stream; the back half stays on the handle's stream
    hipEvent_t ev_front[2];       // set p filled by the front half
    hipEvent_t ev_back[2];        // set p released by the back half
    hipEvent_t ev_chunk[2];       // chunk c & 1 of the pinned staging complete
    // ... and a tail stream with second output-spectra and result buffers: synthesis and copy back of hop h run beside
    // the joint diagonalisation of hop h+1 instead of in front of it
    void* outspec1;               // like outspec, out; hops alternate between the two
    void* out1;
    hipStream_t tail;
    hipEvent_t ev_copied[2];      // result buffer b copied to the host
    void* sig_in;                 // pinned [2][chunk][2][H]
    void* sig_out;                // pinned [2][chunk] hop results (samples [n_out][H] + status words [2][K] each)
    int sig_chunk;                // hops per half of the pinned staging
    // K1 by fast convolution (fir_F > 0): spectra of the zero-padded impulse responses and of the two input histories
    // of the hop, in the front-end precision
    int fir_F;                    // segment length, 0 = direct form
    int fir_np;                   // partitions of the uniformly partitioned form (responses too long for one segment), else 1
    int keep;                     // input samples the histories keep in front of the hop: P - 1, or fir_np H when partitioned
    void* rirspec[2];             // [fir_np][C][fir_F/2 + 1] complex, zone A, zone B
    void* trirspec[2];            // [fir_np][M][fir_F/2 + 1]
    void* xspec;                  // [fir_np][2][fir_F/2 + 1]
    void* xspec_chunk;            // [sig_chunk][2][fir_F/2 + 1]: whole-signal path, the spectra of a staged chunk in one launch
    long hop;                     // hops processed
    long not_converged;           // hops in which some bin hit the sweep cap (status 2)
    // chunked whole-signal path (process_signal_chunked_t): the FRONT half of a whole chunk of hops is three launches (input side,
    // K1, analysis) into linear buffers and one spectra set per hop; the back halves of consecutive hops alternate between two
    // streams with their own filters, output spectra and result buffers.  Everything is allocated at the path's first call.
    int ck_RL;                    // row length of the linear buffers: N - H + sig_chunk H (0 = not set up)
    void* ck_resp[2][4];          // [chunk parity][path]: [C][RL]
    void* ck_tresp[2][2];         // [M][RL]
    void* ck_in[2];               // [2][RL]
    void* ck_X[4];                // [2 sig_chunk][K][C]: set (parity, i) = parity sig_chunk + i
    void* ck_tspec[2];            // [2 sig_chunk][K][M]
    void* ck_inspec;              // [2 sig_chunk][2][K]
    void* ck_w[CK_NB - 1][2];     // [K][nV][L] per zone: filters of back streams 1 ... (back stream 0 uses w, lam)
    void* ck_lam[CK_NB - 1][2];
    void* ck_wall[2];             // [sig_chunk][K][nV][L] per zone: the filters of every hop of a chunk whose joint diagonalisations are ONE launch
    void* ck_lamall[2];           // [sig_chunk][K][L]
    void* ck_spill[CK_NB - 1];    // per-bin scratch of the joint diagonalisation (orders 33..64) for back streams 1 ...: two hops' workgroups for
                                  // the same bin run at once, each parks its state in its own stream's slot (back stream 0 uses d_Lspill)
    hipStream_t ck_back[CK_NB - 1];
    void* ck_pin_in;              // pinned [CK_NS][sig_chunk][2][H]: the host stages chunk c + 1 while chunks c - 1 and c are in flight
    void* ck_pin_out;             // pinned [CK_NS][sig_chunk] hop results
    hipEvent_t ck_done[CK_NS];    // the chunk in staging slot q has been copied back
    void* ck_out;                 // [2 sig_chunk] hop results (samples [n_out][H] + status words [2][K]): one slot per hop of two chunks
    void* ck_ospec;               // [2 sig_chunk][n_out][K] output spectra, likewise
    hipEvent_t ck_backdone[2][CK_NB]; // [chunk parity][back stream]: that stream has read the parity's spectra sets for the last time
    hipEvent_t ck_k3[CK_NB];      // [back stream]: output spectra written (the tail stream starts the synthesis there)
    // work space of apv_stream_get_statistics (R_B, R_D, r, U, w, lam, spill, status), allocated at its first call and kept
    void* stat_ws[10];
    size_t stat_spill_bytes;
    std::vector<hipGraphExec_t> execs;
    std::vector<int32_t> h_status;
};

namespace {

#define SCHK(h, call)                                                                    \
    do {                                                                                 \
        hipError_t _e = (call);                                                          \
        if (_e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

// `count` elements of `esz` bytes, zeroed
template <typename T>
int dalloc(apv_handle* h, T** p, size_t count, size_t esz = sizeof(T)) {
    SCHK(h, hipMalloc((void**)p, esz * (count ? count : 1)));
    SCHK(h, hipMemsetAsync(*p, 0, esz * (count ? count : 1), h->stream));
    return APV_OK;
}

// front-end precision of a streaming handle: cfg.frontend = 0 follows compute_dtype, 1 forces float32, 2 float64
int frontend_f64(const apv_config& c) {
    if (c.frontend == 1) return 0;
    if (c.frontend == 2) return 1;
    return c.compute_dtype == APV_F64;
}

// Uploads go on the handle's stream, like the zero-fill of dalloc before them: that stream does not synchronise with the
// null stream, so a plain hipMemcpy could overtake the pending fill and be wiped by it.
template <typename T>
int upload_as(apv_handle* h, void* dst, const std::vector<double>& src) {
    std::vector<T> tmp(src.begin(), src.end());
    SCHK(h, hipMemcpyAsync(dst, tmp.data(), sizeof(T) * tmp.size(), hipMemcpyHostToDevice, h->stream));
    SCHK(h, hipStreamSynchronize(h->stream));               // tmp goes out of scope
    return APV_OK;
}
int upload(apv_handle* h, int f64, void* dst, const std::vector<double>& src) {
    return f64 ? upload_as<double>(h, dst, src) : upload_as<float>(h, dst, src);
}

size_t wsz(const apv_handle* h) { return h->cfg.out_c128 ? 16 : 8; }
size_t lsz(const apv_handle* h) { return h->cfg.out_c128 ? 8 : 4; }

// bytes of one hop's result: [n_out][H] samples and, behind them, the [2][K] status words
inline size_t hop_out_bytes(const apv_stream* s) { return s->esz * (size_t)s->n_out * s->H; }
inline size_t hop_result_bytes(const apv_stream* s) { return hop_out_bytes(s) + sizeof(int32_t) * 2 * (size_t)s->K; }
inline const int32_t* hop_status_of(const apv_stream* s, const void* result) {
    return reinterpret_cast<const int32_t*>(static_cast<const char*>(result) + hop_out_bytes(s));
}

// path p: signal sig(p) through the RIRs of zone zone(p): AA, AB, BA, BB
inline int path_sig(int p) { return p >> 1; }
inline int path_zone(int p) { return p & 1; }

}  // namespace

void apv_stream_free(apv_handle* h) {
    apv_stream* s = h->st;
    if (!s) return;
    void* bufs[] = {s->rir[0], s->rir[1], s->trir[0], s->trir[1], s->xhist[0][0], s->xhist[0][1], s->xhist[1][0],
                    s->xhist[1][1], s->resp[0], s->resp[1], s->resp[2], s->resp[3], s->tresp[0], s->tresp[1],
                    s->inblk, s->X[0], s->X[1], s->X[2], s->X[3], s->tspec[0], s->tspec[1], s->inspec, s->w[0],
                    s->w[1], s->lam[0], s->lam[1], s->tgt, s->outspec, s->outov, s->out,
                    s->G2, s->G2T, s->Wgt[0], s->Wgt[1], s->rirspec[0], s->rirspec[1], s->trirspec[0], s->trirspec[1], s->xspec, s->xspec_chunk};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (hipGraphExec_t e : s->execs)
        if (e) (void)hipGraphExecDestroy(e);
    if (s->pin_in) (void)hipHostFree(s->pin_in);
    if (s->pin_out) (void)hipHostFree(s->pin_out);
    void* second[] = {s->X1[0], s->X1[1], s->X1[2], s->X1[3], s->tspec1[0], s->tspec1[1], s->inspec1};
    for (void* b : second)
        if (b) (void)hipFree(b);
    for (int p = 0; p < 2; ++p) {
        if (s->ev_front[p]) (void)hipEventDestroy(s->ev_front[p]);
        if (s->ev_back[p]) (void)hipEventDestroy(s->ev_back[p]);
        if (s->ev_chunk[p]) (void)hipEventDestroy(s->ev_chunk[p]);
        if (s->ev_copied[p]) (void)hipEventDestroy(s->ev_copied[p]);
    }
    if (s->out1) (void)hipFree(s->out1);
    if (s->outspec1) (void)hipFree(s->outspec1);
    if (s->front) (void)hipStreamDestroy(s->front);
    if (s->tail) (void)hipStreamDestroy(s->tail);
    if (s->sig_in) (void)hipHostFree(s->sig_in);
    if (s->sig_out) (void)hipHostFree(s->sig_out);
    for (void* b : s->stat_ws)
        if (b) (void)hipFree(b);
    for (int q = 0; q < 2; ++q) {
        for (int p = 0; p < 4; ++p)
            if (s->ck_resp[q][p]) (void)hipFree(s->ck_resp[q][p]);
        for (int z = 0; z < 2; ++z)
            if (s->ck_tresp[q][z]) (void)hipFree(s->ck_tresp[q][z]);
        if (s->ck_in[q]) (void)hipFree(s->ck_in[q]);
        if (s->ck_tspec[q]) (void)hipFree(s->ck_tspec[q]);
        for (int b = 0; b < CK_NB; ++b)
            if (s->ck_backdone[q][b]) (void)hipEventDestroy(s->ck_backdone[q][b]);
    }
    for (int b = 0; b < CK_NB; ++b)
        if (s->ck_k3[b]) (void)hipEventDestroy(s->ck_k3[b]);
    for (int z = 0; z < 2; ++z) {
        if (s->ck_wall[z]) (void)hipFree(s->ck_wall[z]);
        if (s->ck_lamall[z]) (void)hipFree(s->ck_lamall[z]);
    }
    for (int b = 0; b + 1 < CK_NB; ++b) {
        for (int z = 0; z < 2; ++z) {
            if (s->ck_w[b][z]) (void)hipFree(s->ck_w[b][z]);
            if (s->ck_lam[b][z]) (void)hipFree(s->ck_lam[b][z]);
        }
        if (s->ck_spill[b]) (void)hipFree(s->ck_spill[b]);
        if (s->ck_back[b]) (void)hipStreamDestroy(s->ck_back[b]);
    }
    for (int q = 0; q < CK_NS; ++q)
        if (s->ck_done[q]) (void)hipEventDestroy(s->ck_done[q]);
    if (s->ck_pin_in) (void)hipHostFree(s->ck_pin_in);
    if (s->ck_pin_out) (void)hipHostFree(s->ck_pin_out);
    for (int p = 0; p < 4; ++p)
        if (s->ck_X[p]) (void)hipFree(s->ck_X[p]);
    if (s->ck_inspec) (void)hipFree(s->ck_inspec);
    if (s->ck_out) (void)hipFree(s->ck_out);
    if (s->ck_ospec) (void)hipFree(s->ck_ospec);
    delete s;
    h->st = nullptr;
}

// the spectra of one hop: set 0 is the handle's own (state arrays, apv_stream_get_statistics), set 1 exists once
// apv_process_signal has been called
struct HopSpectra {
    void* X[4];
    void* tspec[2];
    void* inspec;
};
static HopSpectra hop_spectra(const apv_stream* s, int set) {
    HopSpectra q;
    for (int p = 0; p < 4; ++p) q.X[p] = set ? s->X1[p] : s->X[p];
    for (int z = 0; z < 2; ++z) q.tspec[z] = set ? s->tspec1[z] : s->tspec[z];
    q.inspec = set ? s->inspec1 : s->inspec;
    return q;
}

// Front half of a hop on stream `st`: pinned hop `pin_src` [2][H] -> input histories, response rings (K1), analysis
// spectra of set `set` (K2, perceptual weighting).  Advances (ring_off, cur) on the host.  Pure enqueue: also used
// under stream capture.
// `set_free` (whole-signal path): event to wait for before the first kernel that writes the spectra set; `xspec_ready`: the
// spectra of this hop's input histories, already formed by apv_launch_fir_chunk_spectra.
static int enqueue_front(apv_handle* h, hipStream_t st, int set, const void* pin_src, hipEvent_t set_free = nullptr,
                         const void* xspec_ready = nullptr) {
    apv_stream* s = h->st;
    const HopSpectra q = hop_spectra(s, set);
    const int N = s->N, H = s->H, K = s->K, L = s->L, M = s->M, C = s->C, P = s->P, f64 = s->f64;
    std::string why;
    // hop -> device, input history and input-block rings
    // the hop [A | B] is read by the input-update kernel straight from the pinned host staging (16 KB over PCIe inside a kernel
    // that has nothing else to do) instead of through a copy node of its own
    const int nxt = s->cur ^ 1;
    // all rings advance by one hop: logical sample n now lives H further on
    s->ring_off = (s->ring_off + H) % N;
    const void* oh[2] = {s->xhist[s->cur][0], s->xhist[s->cur][1]};
    void* nh[2] = {s->xhist[nxt][0], s->xhist[nxt][1]};
    // with the hop's input spectra in hand (whole-signal path) K1 does not wait for the input update: it rides in K1's launch
    const bool ride = xspec_ready != nullptr && s->fir_F > 0 && s->fir_np == 1;
    // (the histories keep s->keep samples in front of the hop: P - 1, more when K1 is partitioned)
    if (!ride) SCHK(h, apv_launch_input_update(f64, s->keep + 1, H, s->pad, N, s->ring_off, oh, nh, pin_src, s->inblk, st));   // histories + input-block rings
    s->cur = nxt;
    // K1: RIR convolution into the response rings (one MFMA launch for all six filter banks)
    if (s->fir_F > 0) {
        const size_t spec_bytes = ((size_t)s->fir_F / 2 + 1) * 2 * s->esz;
        const void* const xs = (xspec_ready && s->fir_np == 1) ? xspec_ready : s->xspec;
        if (s->fir_np > 1)
            SCHK(h, apv_launch_fir_input_spectra_parts(f64, s->fir_F, s->fir_np, s->xhist[s->cur][0], s->xhist[s->cur][1], s->xspec, st));
        else if (!xspec_ready)
            SCHK(h, apv_launch_fir_input_spectra(f64, s->fir_F, s->xhist[s->cur][0], s->xhist[s->cur][1], P - 1 + H, s->xspec, st));
        const void *jh[6], *jx[6];
        void* jr[6];
        int jc[6];
        for (int p = 0; p < 4; ++p) {
            jh[p] = s->rirspec[path_zone(p)]; jx[p] = (const char*)xs + spec_bytes * path_sig(p);
            jr[p] = s->resp[p]; jc[p] = C;
        }
        for (int z = 0; z < 2; ++z) {
            jh[4 + z] = s->trirspec[z]; jx[4 + z] = (const char*)xs + spec_bytes * z;
            jr[4 + z] = s->tresp[z]; jc[4 + z] = M;
        }
        ApvInputUpdate upd{{oh[0], oh[1]}, {nh[0], nh[1]}, pin_src, s->inblk, s->pad};
        SCHK(h, apv_launch_fir_fft_jobs(f64, s->fir_F, 6, jh, jx, jr, jc, P, H, N, s->ring_off, ride ? &upd : nullptr, st, s->fir_np));
    } else if (f64) {
        FirJobsD jobs{};
        for (int p = 0; p < 4; ++p) {
            jobs.rir[p] = (const double*)s->rir[path_zone(p)]; jobs.xh[p] = (const double*)s->xhist[s->cur][path_sig(p)];
            jobs.resp[p] = (double*)s->resp[p]; jobs.C[p] = C;
        }
        for (int z = 0; z < 2; ++z) {
            jobs.rir[4 + z] = (const double*)s->trir[z]; jobs.xh[4 + z] = (const double*)s->xhist[s->cur][z];
            jobs.resp[4 + z] = (double*)s->tresp[z]; jobs.C[4 + z] = M;
        }
        SCHK(h, apv_launch_fir_jobs_f64(jobs, 6, P, H, N, s->ring_off, st));
    } else {
        static const bool valu_fir = (getenv("APV_FIR_VALU") != nullptr);     // A/B switch: direct-form VALU kernel
        if (valu_fir) {
            for (int p = 0; p < 4; ++p)
                SCHK(h, apv_launch_fir_hop(C, P, H, N, s->ring_off, (const float*)s->rir[path_zone(p)],
                                           (const float*)s->xhist[s->cur][path_sig(p)], (float*)s->resp[p], st));
            for (int z = 0; z < 2; ++z)
                SCHK(h, apv_launch_fir_hop(M, P, H, N, s->ring_off, (const float*)s->trir[z], (const float*)s->xhist[s->cur][z],
                                           (float*)s->tresp[z], st));
        } else {
            FirJobs jobs;
            jobs.n = 6;
            for (int p = 0; p < 4; ++p)
                jobs.j[p] = FirJob{(const float*)s->rir[path_zone(p)], (const float*)s->xhist[s->cur][path_sig(p)], (float*)s->resp[p], C};
            for (int z = 0; z < 2; ++z)
                jobs.j[4 + z] = FirJob{(const float*)s->trir[z], (const float*)s->xhist[s->cur][z], (float*)s->tresp[z], M};
            SCHK(h, apv_launch_fir_jobs(jobs, P, H, N, s->ring_off, st));
        }
    }
    // K2: analysis, bin-major output
    const bool runA = s->zones & 1, runB = s->zones & 2;
    if (set_free) SCHK(h, hipStreamWaitEvent(st, set_free, 0));       // histories, rings and K1 above did not need the set
    {
        // every analysis transform of the hop in one launch: the live response paths, both targets, the two inputs
        const void* jx[7];
        void* jspec[7];
        int jch[7], nj = 0;
        long jsc[7], jsk[7];
        for (int p = 0; p < 4; ++p) {
            const bool need = (p < 2) ? runA : runB;       // A->A, A->B feed zone program A; B->A, B->B feed B
            if (!need) continue;
            jx[nj] = s->resp[p]; jspec[nj] = q.X[p]; jch[nj] = C; jsc[nj] = s->xg; jsk[nj] = C; ++nj;      // (xg, C): grouped when xg > 1
        }
        for (int z = 0; z < 2; ++z) { jx[nj] = s->tresp[z]; jspec[nj] = q.tspec[z]; jch[nj] = M; jsc[nj] = 1; jsk[nj] = M; ++nj; }
        jx[nj] = s->inblk; jspec[nj] = q.inspec; jch[nj] = 2; jsc[nj] = K; jsk[nj] = 1; ++nj;
        hipError_t e = apv_launch_stft_analysis_jobs(f64, N, nj, jx, jch, jspec, jsc, jsk, s->ring_off, st, &why);
        if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
    }
    if (s->nch > 0) {
        // weights from the UNWEIGHTED target spectra (apvast.py:205), then spectra x weights (apvast.py:208-209,
        // 258-262): A->A and B->A take zone A's curve, A->B and B->B zone B's
        for (int z = 0; z < 2; ++z)
            SCHK(h, apv_launch_perceptual_weights(f64, K, M, s->nch, q.tspec[z], s->G2, s->G2T, s->Cs, s->Ca, s->Leff, N,
                                                  s->norm_mode, s->Wgt[z], st));
        for (int p = 0; p < 4; ++p) {
            const bool need = (p < 2) ? runA : runB;
            if (need) SCHK(h, apv_launch_scale_spectra(f64, K, C, L, q.X[p], s->Wgt[path_zone(p)], st));
        }
        for (int z = 0; z < 2; ++z) SCHK(h, apv_launch_scale_spectra(f64, K, M, 1, q.tspec[z], s->Wgt[z], st));
    }
    return APV_OK;
}

// How the whole-signal path runs the back half: which output-spectra / result buffers, an event to record once the
// spectra set has been read for the last time (after K3), and a stream of its own for synthesis and copy back, which
// start at that event.  The per-hop path takes the defaults.
struct BackSchedule {
    int yield_issue = 0;      // GevdParams::yield_issue
    int obuf = 0;
    hipEvent_t spectra_free = nullptr;
    hipStream_t tail_stream = nullptr;
    hipEvent_t copied = nullptr;
    // chunked whole-signal path: the hop's own result and output-spectra slots (then `obuf` is not used) and no copy back per
    // hop (one copy per chunk, by the caller)
    void* result = nullptr;
    void* ospec = nullptr;
    bool no_copy = false;
    void* lspill = nullptr;   // GevdParams::Lspill of this launch (nullptr: the handle's d_Lspill)
    bool skip_gevd = false;   // the filters are there already: the chunk's joint diagonalisations were one launch (enqueue_gevd_hops)
};

// Back half of a hop on stream `st`: spectra of set `set` -> per-bin filters (K5'-K10), output spectra (K3), synthesis
// and overlap-add (K4); the emitted samples [n_out][H] and, behind them, the status words [2][K] land in pinned `pin_dst`.
static int enqueue_back(apv_handle* h, hipStream_t st, const HopSpectra& q, void* const* wz, void* const* lamz, void* pin_dst,
                        const BackSchedule& sch = BackSchedule()) {
    apv_stream* s = h->st;
    char* const obuf = static_cast<char*>(sch.result ? sch.result : (sch.obuf ? s->out1 : s->out));
    char* const ospec = static_cast<char*>(sch.ospec ? sch.ospec : (sch.obuf ? s->outspec1 : s->outspec));
    int32_t* const ostatus[2] = {reinterpret_cast<int32_t*>(obuf + hop_out_bytes(s)),
                                 reinterpret_cast<int32_t*>(obuf + hop_out_bytes(s)) + s->K};
    const int N = s->N, H = s->H, K = s->K, L = s->L, f64 = s->f64;
    const size_t e2 = 2 * s->esz;
    const bool runA = s->zones & 1, runB = s->zones & 2;
    std::string why;
    // per-bin update per zone program: A: bright A->A, dark A->B, target A;  B: bright B->B, dark B->A, target B
    int oc = 0;     // output channel cursor
    if (!sch.skip_gevd) {
        // both zone programs go out in ONE launch (blockIdx.y = zone): K = N/2+1 bins alone cannot fill the chip
        GevdParams p = apv_base_params(h);
        const int first = runA ? 0 : 1;
        p.x_c128 = f64;
        p.x_group = s->xg;
        p.XB = first ? q.X[3] : q.X[0];
        p.XD = first ? q.X[2] : q.X[1];
        p.d = q.tspec[first];
        p.w = wz[first];
        p.lam = lamz[first];
        p.status = ostatus[first];
        p.n_zones = (runA && runB) ? 2 : 1;
        p.yield_issue = sch.yield_issue;
        if (sch.lspill) p.Lspill = sch.lspill;
        if (p.n_zones == 2) {
            p.XB1 = q.X[3]; p.XD1 = q.X[2]; p.d1 = q.tspec[1];
            p.w1 = wz[1]; p.lam1 = lamz[1]; p.status1 = ostatus[1];
        }
        hipError_t e = apv_launch_gevd(p, h->cfg.compute_dtype, true, st, &why);
        if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
    }
    {
        // K3: output spectra in one launch: each live zone's nV*L filtered channels, then the target paths A_t, B_t
        const void* jin[4];
        const void* jw[4];
        const void* jt[4];
        void* jout[4];
        int jf[4], jtg[4], nj = 0;
        for (int z = 0; z < 2; ++z) {
            if (!(z ? runB : runA)) continue;
            jin[nj] = (const char*)q.inspec + (size_t)z * K * e2; jw[nj] = wz[z]; jt[nj] = nullptr;
            jout[nj] = ospec + (size_t)oc * K * e2;
            jf[nj] = s->nV * L; jtg[nj] = 0; ++nj;
            oc += s->nV * L;
        }
        for (int z = 0; z < 2; ++z) {
            jin[nj] = (const char*)q.inspec + (size_t)z * K * e2; jw[nj] = nullptr; jt[nj] = s->tgt;
            jout[nj] = ospec + (size_t)oc * K * e2;
            jf[nj] = 0; jtg[nj] = L; ++nj;
            oc += L;
        }
        SCHK(h, apv_launch_apply_jobs(K, nj, jin, jw, jt, jout, jf, jtg, h->cfg.out_c128, f64, st));
    }
    if (sch.spectra_free) SCHK(h, hipEventRecord(sch.spectra_free, st));
    hipStream_t ts = st;
    if (sch.tail_stream) {
        ts = sch.tail_stream;
        SCHK(h, hipStreamWaitEvent(ts, sch.spectra_free, 0));
    }
    // K4: synthesis + overlap-add + emit
    {
        hipError_t e = apv_launch_synthesis(f64, N, H, s->n_out, ospec, K, 1, s->outov, obuf, ts, &why, s->out_group);
        if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
    }
    if (sch.no_copy) return APV_OK;
    SCHK(h, hipMemcpyAsync(pin_dst, obuf, hop_result_bytes(s), hipMemcpyDeviceToHost, ts));         // samples + status: one copy
    if (sch.tail_stream) SCHK(h, hipEventRecord(sch.copied, ts));
    return APV_OK;
}

// everything one hop puts on the handle's stream, from the pinned input staging to the pinned output staging
static int enqueue_hop(apv_handle* h) {
    apv_stream* s = h->st;
    int rc = enqueue_front(h, h->stream, 0, s->pin_in);
    if (rc != APV_OK) return rc;
    return enqueue_back(h, h->stream, hop_spectra(s, 0), s->w, s->lam, s->pin_out);
}

// run one hop whose input is already in the pinned staging; the output is left in the pinned staging
static int run_hop(apv_handle* h) {
    apv_stream* s = h->st;
    hipStream_t st = h->stream;
    const int H = s->H;
    if (s->period > 0) {
        // replay the captured launch sequence of this phase; capture it the first time the phase comes up
        const int phase = (int)(s->hop % s->period);
        const int ring_before = s->ring_off, cur_before = s->cur;
        if (!s->execs[phase]) {
            hipGraph_t graph = nullptr;
            SCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = enqueue_hop(h);
            hipError_t ce = hipStreamEndCapture(st, &graph);
            if (rc != APV_OK || ce != hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                s->ring_off = ring_before;
                s->cur = cur_before;
                s->period = 0;                       // launch eagerly from now on
                rc = enqueue_hop(h);
                if (rc != APV_OK) return rc;
            } else {
                SCHK(h, hipGraphInstantiate(&s->execs[phase], graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                SCHK(h, hipGraphLaunch(s->execs[phase], st));
            }
        } else {
            SCHK(h, hipGraphLaunch(s->execs[phase], st));
            // the host-side cursor moves exactly as enqueue_hop moves it
            s->cur ^= 1;
            s->ring_off = (s->ring_off + H) % s->N;
        }
    } else {
        int rc = enqueue_hop(h);
        if (rc != APV_OK) return rc;
    }
    s->hop++;
    SCHK(h, hipStreamSynchronize(st));
    return APV_OK;
}

// scan the per-bin status words of the hop: 1 = not positive definite (error, apvast.py:21-24), 2 = sweep cap reached
static int scan_hop_status(apv_handle* h, const int32_t* stat, long hop) {
    apv_stream* s = h->st;
    const int K = s->K;
    const bool runA = s->zones & 1, runB = s->zones & 2;
    int slow_zone = -1, slow_bin = -1;
    for (int z = 0; z < 2; ++z) {
        if (!(z ? runB : runA)) continue;
        for (int k = 0; k < K; ++k) {
            const int v = stat[(size_t)z * K + k];
            if (v == 1) {
                char buf[112];
                std::snprintf(buf, sizeof(buf), "Matrix is not positive definite (zone %c, bin %d, hop %ld)", z ? 'B' : 'A', k, hop);
                return apv_fail(h, APV_ERR_NOT_PD, buf);
            }
            if (v == 2 && slow_zone < 0) { slow_zone = z; slow_bin = k; }
        }
    }
    if (slow_zone >= 0) {
        s->not_converged++;
        char buf[128];
        std::snprintf(buf, sizeof(buf), "eigen-iteration did not converge (zone %c, bin %d, hop %ld); the outputs of this hop were written",
                      slow_zone ? 'B' : 'A', slow_bin, hop);
        return apv_fail(h, APV_ERR_NO_CONVERGE, buf);
    }
    return APV_OK;
}

template <typename TI>
static int process_block_t(apv_handle* h, const TI* h_in_A, const TI* h_in_B, TI* h_out) {
    if (!h || !h_in_A || !h_in_B || !h_out) return apv_fail(h, APV_ERR_ARG, "null argument");
    apv_stream* s = h->st;
    if (!s) return apv_fail(h, APV_ERR_ARG, "apv_stream_init has not been called");
    SCHK(h, hipSetDevice(h->device));
    const int H = s->H;
    const size_t nout = (size_t)s->n_out * H;
    if (s->f64) {
        double* pi = (double*)s->pin_in;
        for (int i = 0; i < H; ++i) { pi[i] = (double)h_in_A[i]; pi[H + i] = (double)h_in_B[i]; }
    } else {
        float* pi = (float*)s->pin_in;
        for (int i = 0; i < H; ++i) { pi[i] = (float)h_in_A[i]; pi[H + i] = (float)h_in_B[i]; }
    }
    int rc = run_hop(h);
    if (rc != APV_OK) return rc;
    if (s->f64) {
        const double* po = (const double*)s->pin_out;
        if (sizeof(TI) == sizeof(double)) std::memcpy(h_out, po, nout * sizeof(double));
        else for (size_t i = 0; i < nout; ++i) h_out[i] = (TI)po[i];
    } else {
        const float* po = (const float*)s->pin_out;
        if (sizeof(TI) == sizeof(float)) std::memcpy(h_out, po, nout * sizeof(float));
        else for (size_t i = 0; i < nout; ++i) h_out[i] = (TI)po[i];
    }
    return scan_hop_status(h, hop_status_of(s, s->pin_out), s->hop - 1);
}

// what the whole-signal path needs beyond the per-hop path; allocated at its first call
static int signal_prepare(apv_handle* h) {
    apv_stream* s = h->st;
    if (s->sig_chunk > 0) return APV_OK;
    const size_t K = s->K, C = s->C, M = s->M, H = s->H, e1 = s->esz, e2 = 2 * s->esz;
    // hops per chunk of the whole-signal path (APV_SIGNAL_CHUNK: tuning aid, 4..64)
    static const int chunk_env = getenv("APV_SIGNAL_CHUNK") ? atoi(getenv("APV_SIGNAL_CHUNK")) : 0;
    const int chunk = (chunk_env >= 4 && chunk_env <= 64) ? chunk_env : 16;
    int rc;
    for (int p = 0; p < 4; ++p)
        if (!s->X1[p] && (rc = dalloc(h, &s->X1[p], (size_t)s->Kp * C, e2))) return rc;
    for (int z = 0; z < 2; ++z)
        if (!s->tspec1[z] && (rc = dalloc(h, &s->tspec1[z], K * M, e2))) return rc;
    if (!s->inspec1 && (rc = dalloc(h, &s->inspec1, 2 * K, e2))) return rc;
    if (!s->out1 && (rc = dalloc(h, &s->out1, hop_result_bytes(s), 1))) return rc;
    if (!s->outspec1 && (rc = dalloc(h, &s->outspec1, (size_t)s->n_out * K, e2))) return rc;
    if (s->fir_F > 0 && !s->xspec_chunk && (rc = dalloc(h, &s->xspec_chunk, (size_t)chunk * 2 * (s->fir_F / 2 + 1), e2))) return rc;
    // Front and tail at the DEFAULT stream priority.  (Rounds 2-4 ran them at the highest: their kernels are short or bandwidth-shaped
    // and everything downstream waits for them, and beside per-hop joint diagonalisations that is worth 3-6 % in a fresh process
    // -- but with priority streams the rate of the hop-by-hop schedule depends on how many streams the process created before:
    // 0.069 / 0.102 / 0.092 ms per hop at cfg3 after 0 / 3 / 5 earlier streams, against 0.074 / 0.071 / 0.074 at the default
    // priority (profiles/r04/cfg3_front_prio.txt).  The schedule with one joint-diagonalisation launch per chunk measures the same
    // either way.  APV_CK_FRONT_PRIO=1 asks for the highest priority again: A/B switch.)
    int prio_least = 0, prio_greatest = 0;
    SCHK(h, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    static const bool front_high = getenv("APV_CK_FRONT_PRIO") != nullptr && atoi(getenv("APV_CK_FRONT_PRIO")) == 1;
    if (!front_high) prio_greatest = 0;
    if (!s->front) SCHK(h, hipStreamCreateWithPriority(&s->front, hipStreamNonBlocking, prio_greatest));
    if (!s->tail) SCHK(h, hipStreamCreateWithPriority(&s->tail, hipStreamNonBlocking, prio_greatest));
    for (int p = 0; p < 2; ++p) {
        if (!s->ev_copied[p]) SCHK(h, hipEventCreateWithFlags(&s->ev_copied[p], hipEventDisableTiming));
        if (!s->ev_front[p]) SCHK(h, hipEventCreateWithFlags(&s->ev_front[p], hipEventDisableTiming));
        if (!s->ev_back[p]) SCHK(h, hipEventCreateWithFlags(&s->ev_back[p], hipEventDisableTiming));
        if (!s->ev_chunk[p]) SCHK(h, hipEventCreateWithFlags(&s->ev_chunk[p], hipEventDisableTiming));
    }
    // two chunks of pinned staging: the host converts chunk c-1 while the device runs chunk c
    if (!s->sig_in) SCHK(h, hipHostMalloc(&s->sig_in, e1 * 2 * chunk * 2 * H, hipHostMallocDefault));
    if (!s->sig_out) {
        SCHK(h, hipHostMalloc(&s->sig_out, 2 * chunk * hop_result_bytes(s), hipHostMallocDefault));
        std::memset(s->sig_out, 0, 2 * chunk * hop_result_bytes(s));
    }
    SCHK(h, hipStreamSynchronize(h->stream));               // the zero-fills above
    s->sig_chunk = chunk;
    return APV_OK;
}

// n_hops consecutive hops in one call.  Per hop the kernels, their order and their operands are those of
// apv_process_block, so the samples returned are the same bit for bit; what changes is the schedule: the front half of
// hop h+1 (nothing in it depends on hop h's filters) runs on a second stream beside the back half of hop h, the spectra
// alternating between two sets, and the host never waits for a hop: it stages and converts one chunk of hops while the
// device works on the next.
template <typename TI>
static int process_signal_chunked_t(apv_handle* h, int n_hops, const TI* h_in_A, const TI* h_in_B, TI* h_out);

template <typename TI>
static int process_signal_t(apv_handle* h, int n_hops, const TI* h_in_A, const TI* h_in_B, TI* h_out) {
    if (!h || !h_in_A || !h_in_B || !h_out) return apv_fail(h, APV_ERR_ARG, "null argument");
    apv_stream* s = h->st;
    if (!s) return apv_fail(h, APV_ERR_ARG, "apv_stream_init has not been called");
    if (n_hops < 0) return apv_fail(h, APV_ERR_ARG, "n_hops must be >= 0");
    if (n_hops == 0) return APV_OK;
    // responses of 64 taps or more (K1 by fast convolution): a chunk of hops per launch, two back streams (below); short
    // responses, APV_FIR_DIRECT and APV_SIGNAL_PER_HOP (A/B switch) keep the hop-by-hop pipeline of this function
    static const bool per_hop = getenv("APV_SIGNAL_PER_HOP") != nullptr;
    if (s->fir_F > 0 && s->fir_np == 1 && !per_hop) return process_signal_chunked_t<TI>(h, n_hops, h_in_A, h_in_B, h_out);
    SCHK(h, hipSetDevice(h->device));
    int rc = signal_prepare(h);
    if (rc != APV_OK) return rc;
    const int H = s->H, K = s->K, chunk = s->sig_chunk;
    const size_t e1 = s->esz, nout = (size_t)s->n_out * H;
    hipStream_t back = h->stream;
    auto drain = [&]() { (void)hipStreamSynchronize(s->front); (void)hipStreamSynchronize(back); (void)hipStreamSynchronize(s->tail); };
    int set = 0, last_set = 0;
    bool released[2] = {true, true};                         // nothing reads either set yet
    bool out_idle[2] = {true, true};                         // no copy of either result buffer pending
    int worst = APV_OK;                                      // first APV_ERR_NO_CONVERGE, overridden by APV_ERR_NOT_PD
    std::string worst_msg;
    const int n_chunks = (n_hops + chunk - 1) / chunk;
    const long hop_first = s->hop;
    // chunk c of the signal lives in half c & 1 of the pinned staging
    auto stage_in = [&](int c) {
        const int base = c * chunk, nc = std::min(chunk, n_hops - base);
        for (int i = 0; i < nc; ++i) {
            const TI* a = h_in_A + (size_t)(base + i) * H;
            const TI* b = h_in_B + (size_t)(base + i) * H;
            const size_t slot = ((size_t)(c & 1) * chunk + i) * 2 * H;
            if (s->f64) {
                double* pi = (double*)s->sig_in + slot;
                for (int t = 0; t < H; ++t) { pi[t] = (double)a[t]; pi[H + t] = (double)b[t]; }
            } else {
                float* pi = (float*)s->sig_in + slot;
                for (int t = 0; t < H; ++t) { pi[t] = (float)a[t]; pi[H + t] = (float)b[t]; }
            }
        }
    };
    auto collect = [&](int c) -> int {
        const int base = c * chunk, nc = std::min(chunk, n_hops - base);
        hipError_t e = hipEventSynchronize(s->ev_chunk[c & 1]);
        if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string("apv_process_signal: ") + hipGetErrorString(e));
        // channel-major: h_out [n_hops][n_out][H].  Sample-major: h_out [n_out / L][n_hops * H][L], i.e. the hop's group g, a
        // contiguous [H][L] block of the result, lands behind the same group of the hop before it (one straight copy per group)
        const size_t ngrp = s->out_group > 0 ? (size_t)s->n_out / s->out_group : 1, gsz = nout / ngrp;
        for (int i = 0; i < nc; ++i) {
            const char* res = (const char*)s->sig_out + ((size_t)(c & 1) * chunk + i) * hop_result_bytes(s);
            for (size_t g = 0; g < ngrp; ++g) {
                TI* dst = s->out_group > 0 ? h_out + (g * (size_t)n_hops + (size_t)(base + i)) * gsz : h_out + (size_t)(base + i) * nout;
                if (s->f64) {
                    const double* po = (const double*)res + g * gsz;
                    if (sizeof(TI) == sizeof(double)) std::memcpy(dst, po, gsz * sizeof(double));
                    else for (size_t j = 0; j < gsz; ++j) dst[j] = (TI)po[j];
                } else {
                    const float* po = (const float*)res + g * gsz;
                    if (sizeof(TI) == sizeof(float)) std::memcpy(dst, po, gsz * sizeof(float));
                    else for (size_t j = 0; j < gsz; ++j) dst[j] = (TI)po[j];
                }
            }
            const int r = scan_hop_status(h, hop_status_of(s, res), hop_first + base + i);
            if (r == APV_ERR_NOT_PD && worst != APV_ERR_NOT_PD) { worst = r; worst_msg = h->err; }
            if (r == APV_ERR_NO_CONVERGE && worst == APV_OK) { worst = r; worst_msg = h->err; }
        }
        return APV_OK;
    };
    int c_done = 0;                                          // chunks collected
    // Every failure after the first enqueue leaves through here: all three streams are drained (kernels may still be reading the
    // pinned staging and writing the result buffers), and the message says how far the call got -- the rings, histories and
    // the hop counter have advanced by the hops ENQUEUED, of which only the hops DELIVERED reached h_out.
    auto bail = [&](int code, const std::string& msg) {
        drain();
        char buf[192];
        const long enq = s->hop - hop_first;
        const long deliv = std::min<long>((long)c_done * chunk, enq);
        std::snprintf(buf, sizeof(buf), " [apv_process_signal: %ld of %d hops delivered, stream state advanced by %ld hops: restore it "
                      "with apv_set_state or re-initialise before continuing]", deliv, n_hops, enq);
        return apv_fail(h, code, msg + buf);
    };
    auto hipbail = [&](hipError_t e) { return bail(APV_ERR_HIP, std::string("apv_process_signal: ") + hipGetErrorString(e)); };
    for (int c = 0; c < n_chunks && worst != APV_ERR_NOT_PD; ++c) {
        const int base = c * chunk, nc = std::min(chunk, n_hops - base);
        stage_in(c);                                         // this half was collected when chunk c-2's event came in
        const size_t xs_bytes = s->fir_F > 0 ? ((size_t)s->fir_F / 2 + 1) * 2 * e1 * 2 : 0;      // one hop's two spectra
        if (s->fir_F > 0 && s->fir_np == 1) {
            // the input spectra K1 starts from, for every hop of the chunk at once: they depend on the staged samples and on
            // the histories as the previous chunk left them, on nothing of this chunk's processing
            hipError_t e = apv_launch_fir_chunk_spectra(s->f64, s->fir_F, s->P, H, nc, s->xhist[s->cur][0], s->xhist[s->cur][1],
                                                        (const char*)s->sig_in + (size_t)(c & 1) * chunk * 2 * H * e1, s->xspec_chunk,
                                                        s->front);
            if (e != hipSuccess) return hipbail(e);
        }
        for (int i = 0; i < nc; ++i) {
            const size_t slot = (size_t)(c & 1) * chunk + i;
            hipError_t e = hipSuccess;
            {
                // hop h-2 has to be done with this set before the analysis transforms write it
                rc = enqueue_front(h, s->front, set, (const char*)s->sig_in + slot * 2 * H * e1, released[set] ? nullptr : s->ev_back[set],
                                   (s->fir_F > 0 && s->fir_np == 1) ? (const char*)s->xspec_chunk + (size_t)i * xs_bytes : nullptr);
                if (rc != APV_OK) return bail(rc, h->err);
                e = hipEventRecord(s->ev_front[set], s->front);
            }
            if (e == hipSuccess) e = hipStreamWaitEvent(back, s->ev_front[set], 0);
            // the output buffers follow the set; hop h-2's synthesis and copy have to be through before this hop writes them
            if (e == hipSuccess && !out_idle[set]) e = hipStreamWaitEvent(back, s->ev_copied[set], 0);
            if (e == hipSuccess) {
                BackSchedule sch;
                sch.obuf = set;
                sch.spectra_free = s->ev_back[set];          // recorded after K3, the set's last reader
                sch.tail_stream = s->tail;
                sch.copied = s->ev_copied[set];
                rc = enqueue_back(h, back, hop_spectra(s, set), s->w, s->lam, (char*)s->sig_out + slot * hop_result_bytes(s), sch);
                if (rc != APV_OK) return bail(rc, h->err);
                out_idle[set] = false;
            }
            if (e != hipSuccess) return hipbail(e);
            released[set] = false;
            last_set = set;
            set ^= 1;
            s->hop++;
        }
        {
            const hipError_t e = hipEventRecord(s->ev_chunk[c & 1], s->tail);   // every front and back half is upstream of some copy
            if (e != hipSuccess) return hipbail(e);
        }
        if (c > 0) {
            if ((rc = collect(c - 1)) != APV_OK) return bail(rc, h->err);
            c_done = c;
        }
    }
    // the chunks still in flight (one, or none if a hop was not positive definite in the last one collected)
    for (int c = c_done; c < n_chunks && (size_t)c * chunk < (size_t)(s->hop - hop_first); ++c) {
        if ((rc = collect(c)) != APV_OK) return bail(rc, h->err);
        c_done = c + 1;
    }
    if (last_set == 1) {
        // the state arrays and apv_process_block live in set 0: bring the last hop's spectra there
        const size_t C = s->C, M = s->M, e2 = 2 * e1;
        hipError_t e = hipSuccess;
        for (int p = 0; p < 4 && e == hipSuccess; ++p) e = hipMemcpyAsync(s->X[p], s->X1[p], (size_t)s->Kp * C * e2, hipMemcpyDeviceToDevice, back);
        for (int z = 0; z < 2 && e == hipSuccess; ++z) e = hipMemcpyAsync(s->tspec[z], s->tspec1[z], (size_t)K * M * e2, hipMemcpyDeviceToDevice, back);
        if (e == hipSuccess) e = hipMemcpyAsync(s->inspec, s->inspec1, (size_t)2 * K * e2, hipMemcpyDeviceToDevice, back);
        if (e != hipSuccess) return hipbail(e);
    }
    {
        hipError_t e = hipStreamSynchronize(s->front);
        if (e == hipSuccess) e = hipStreamSynchronize(back);
        if (e == hipSuccess) e = hipStreamSynchronize(s->tail);
        if (e != hipSuccess) return hipbail(e);
    }
    if (worst == APV_ERR_NOT_PD) return bail(worst, worst_msg);     // the hops behind the failing one were not run
    if (worst != APV_OK) return apv_fail(h, worst, worst_msg);      // APV_ERR_NO_CONVERGE: every hop ran, every output is written
    return APV_OK;
}

// what the chunked whole-signal path needs beyond signal_prepare(); allocated at its first call
static int chunk_prepare(apv_handle* h) {
    apv_stream* s = h->st;
    if (s->ck_RL > 0) return APV_OK;
    int rc = signal_prepare(h);
    if (rc != APV_OK) return rc;
    const size_t K = s->K, C = s->C, M = s->M, L = s->L, e1 = s->esz, e2 = 2 * s->esz;
    const int chunk = s->sig_chunk;
    const size_t RL = (size_t)s->N - s->H + (size_t)chunk * s->H;
    for (int q = 0; q < 2; ++q) {
        for (int p = 0; p < 4; ++p)
            if ((rc = dalloc(h, &s->ck_resp[q][p], C * RL, e1))) return rc;
        for (int z = 0; z < 2; ++z)
            if ((rc = dalloc(h, &s->ck_tresp[q][z], M * RL, e1))) return rc;
        if ((rc = dalloc(h, &s->ck_in[q], 2 * RL, e1))) return rc;
        if ((rc = dalloc(h, &s->ck_tspec[q], (size_t)2 * chunk * K * M, e2))) return rc;
        for (int b = 0; b < CK_NB; ++b) SCHK(h, hipEventCreateWithFlags(&s->ck_backdone[q][b], hipEventDisableTiming));
    }
    for (int b = 0; b < CK_NB; ++b) SCHK(h, hipEventCreateWithFlags(&s->ck_k3[b], hipEventDisableTiming));
    for (int z = 0; z < 2; ++z) {
        if ((rc = dalloc(h, &s->ck_wall[z], (size_t)chunk * K * s->nV * L, wsz(h)))) return rc;
        if ((rc = dalloc(h, &s->ck_lamall[z], (size_t)chunk * K * L, lsz(h)))) return rc;
    }
    for (int b = 0; b + 1 < CK_NB; ++b) {
        for (int z = 0; z < 2; ++z) {
            if ((rc = dalloc(h, &s->ck_w[b][z], K * s->nV * L, wsz(h)))) return rc;
            if ((rc = dalloc(h, &s->ck_lam[b][z], K * L, lsz(h)))) return rc;
        }
        {
            const apv_config& c = h->cfg;
            const size_t spill = apv_gevd_spill_bytes((int)L, (int)K, c.compute_dtype, c.reg_mode, c.reg_bright, c.sweep_tol2, c.n_zones == 3 ? 2 : 1);
            if (spill > 0 && (rc = dalloc(h, &s->ck_spill[b], spill, 1))) return rc;
        }
        // (the back streams themselves are made by process_signal_chunked_t, and only where the hop-by-hop schedule runs: the
        // runtime deals a process's streams over four hardware queues, and a stream that exists takes part whether it is used or not)
    }
    for (int q = 0; q < CK_NS; ++q) SCHK(h, hipEventCreateWithFlags(&s->ck_done[q], hipEventDisableTiming));
    SCHK(h, hipHostMalloc(&s->ck_pin_in, e1 * CK_NS * chunk * 2 * s->H, hipHostMallocDefault));
    SCHK(h, hipHostMalloc(&s->ck_pin_out, (size_t)CK_NS * chunk * hop_result_bytes(s), hipHostMallocDefault));
    std::memset(s->ck_pin_out, 0, (size_t)CK_NS * chunk * hop_result_bytes(s));
    for (int p = 0; p < 4; ++p)
        if ((rc = dalloc(h, &s->ck_X[p], (size_t)2 * chunk * s->Kp * C, e2))) return rc;
    if ((rc = dalloc(h, &s->ck_inspec, (size_t)2 * chunk * 2 * K, e2))) return rc;
    if ((rc = dalloc(h, &s->ck_out, (size_t)2 * chunk * hop_result_bytes(s), 1))) return rc;
    if ((rc = dalloc(h, &s->ck_ospec, (size_t)2 * chunk * s->n_out * K, e2))) return rc;
    SCHK(h, hipStreamSynchronize(h->stream));               // the zero-fills above
    s->ck_RL = (int)RL;
    return APV_OK;
}

// n_hops consecutive hops in one call, a CHUNK of hops at a time (responses of >= 64 taps, i.e. K1 by fast convolution).
//
// Nothing in the front half of a hop (input histories, K1, analysis transforms, perceptual weighting) depends on any hop's
// filters, and the whole signal is in hand: the front halves of the 16 hops of a chunk are therefore THREE launches -- the input
// side (hops into the input-block buffers, histories as after the chunk), K1 for every (hop, channel), the analysis transform
// of every (hop, channel) -- into linear buffers [channel][N - H + 16 H] whose head is the tail of the previous chunk's, and
// into one spectra set per hop.  Each (hop, channel) is the workgroup the per-hop launch would have run, on the same operands in
// the same order: the samples returned are those of apv_process_block bit for bit.  The back halves (joint diagonalisation,
// output spectra) of consecutive hops are independent of each other and alternate between two streams with their own filters,
// output spectra and result buffers, so two hops' diagonalisations share the chip (four waves per SIMD instead of two);
// synthesis, overlap-add and the copy back stay in hop order on the tail stream.  The host stages chunk c + 1 and collects
// chunk c - 1 while the device runs chunk c.  On return the rings, histories, spectra, filters and counters are what n_hops
// calls of apv_process_block would have left (apv_get_state, the per-hop entry points and their captured graphs carry on).
template <typename TI>
static int process_signal_chunked_t(apv_handle* h, int n_hops, const TI* h_in_A, const TI* h_in_B, TI* h_out) {
    apv_stream* s = h->st;
    SCHK(h, hipSetDevice(h->device));
    int rc = chunk_prepare(h);
    if (rc != APV_OK) return rc;
    const int N = s->N, H = s->H, K = s->K, L = s->L, M = s->M, C = s->C, P = s->P, f64 = s->f64, chunk = s->sig_chunk, RL = s->ck_RL;
    const size_t e1 = s->esz, e2 = 2 * s->esz, nout = (size_t)s->n_out * H;
    const bool runA = s->zones & 1, runB = s->zones & 2;
    // the chunk's joint diagonalisations as one launch where the kernel that will run takes several hops (order 16, absolute loading,
    // no diagnostics); APV_SIGNAL_BATCHED=0: hop by hop on the back streams, as before (A/B switch)
    bool batched = false;
    {
        static const bool want = getenv("APV_SIGNAL_BATCHED") == nullptr || atoi(getenv("APV_SIGNAL_BATCHED")) != 0;
        GevdParams probe = apv_base_params(h);
        probe.x_c128 = f64;
        probe.x_group = s->xg;
        probe.n_hops = 2;
        static const bool force_generic = getenv("APV_FORCE_GENERIC") != nullptr;
        batched = want && !force_generic && apv_gevd16m_takes_hops(probe, h->cfg.compute_dtype, true);
    }
    if (!batched) {
        // experiment (APV_CK_BACK_PRIO): -1 lowest stream priority, 1 highest, else the default
        static const int want = getenv("APV_CK_BACK_PRIO") ? atoi(getenv("APV_CK_BACK_PRIO")) : 0;
        int lo = 0, hi = 0;
        SCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (int b = 0; b + 1 < CK_NB; ++b) {
            if (s->ck_back[b]) continue;
            if (want == 0) SCHK(h, hipStreamCreateWithFlags(&s->ck_back[b], hipStreamNonBlocking));
            else SCHK(h, hipStreamCreateWithPriority(&s->ck_back[b], hipStreamNonBlocking, want < 0 ? lo : hi));
        }
    }
    hipStream_t bs[CK_NB];
    void* wset[CK_NB][2];
    void* lset[CK_NB][2];
    for (int b = 0; b < CK_NB; ++b) {
        bs[b] = (b && s->ck_back[b - 1]) ? s->ck_back[b - 1] : h->stream;
        for (int z = 0; z < 2; ++z) {
            wset[b][z] = b ? s->ck_w[b - 1][z] : s->w[z];
            lset[b][z] = b ? s->ck_lam[b - 1][z] : s->lam[z];
        }
    }
    auto drain = [&]() {
        (void)hipStreamSynchronize(s->front);
        for (int b = 0; b < CK_NB; ++b) (void)hipStreamSynchronize(bs[b]);
        (void)hipStreamSynchronize(s->tail);
    };
    const int n_chunks = (n_hops + chunk - 1) / chunk;
    const long hop_first = s->hop;
    const int ring_first = s->ring_off, cur_first = s->cur;
    static const bool timing = getenv("APV_SIGNAL_TIMING") != nullptr;      // host-side seconds per phase, to stderr
    double t_stage = 0, t_enq = 0, t_wait = 0, t_copy = 0;
    std::vector<hipEvent_t> tev;                             // timing aid: [chunk][front start, front end, back0 start, back end 0, back end 1, copied]
    auto tmark = [&](hipStream_t st) { if (timing) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); tev.push_back(e); } };
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    int worst = APV_OK;
    std::string worst_msg;
    int c_done = 0;
    auto bail = [&](int code, const std::string& msg) {
        drain();
        char buf[192];
        const long enq = s->hop - hop_first;
        const long deliv = std::min<long>((long)c_done * chunk, enq);
        std::snprintf(buf, sizeof(buf), " [apv_process_signal: %ld of %d hops delivered, stream state advanced by %ld hops: restore it "
                      "with apv_set_state or re-initialise before continuing]", deliv, n_hops, enq);
        return apv_fail(h, code, msg + buf);
    };
    auto hipbail = [&](hipError_t e) { return bail(APV_ERR_HIP, std::string("apv_process_signal: ") + hipGetErrorString(e)); };
#define CK(call) do { hipError_t _e = (call); if (_e != hipSuccess) return hipbail(_e); } while (0)
    auto stage_in = [&](int c) {
        const int base = c * chunk, nc = std::min(chunk, n_hops - base);
        for (int i = 0; i < nc; ++i) {
            const TI* a = h_in_A + (size_t)(base + i) * H;
            const TI* b = h_in_B + (size_t)(base + i) * H;
            const size_t slot = ((size_t)(c % CK_NS) * chunk + i) * 2 * H;
            if (s->f64) {
                double* pi = (double*)s->ck_pin_in + slot;
                for (int t = 0; t < H; ++t) { pi[t] = (double)a[t]; pi[H + t] = (double)b[t]; }
            } else {
                float* pi = (float*)s->ck_pin_in + slot;
                for (int t = 0; t < H; ++t) { pi[t] = (float)a[t]; pi[H + t] = (float)b[t]; }
            }
        }
    };
    auto collect = [&](int c) -> int {
        const int base = c * chunk, nc = std::min(chunk, n_hops - base);
        double tc0 = now();
        hipError_t e = hipEventSynchronize(s->ck_done[c % CK_NS]);
        if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string("apv_process_signal: ") + hipGetErrorString(e));
        t_wait += now() - tc0; tc0 = now();
        struct Acc { double& t; double t0; std::function<double()> f; ~Acc() { t += f() - t0; } } acc{t_copy, tc0, now};
        const size_t ngrp = s->out_group > 0 ? (size_t)s->n_out / s->out_group : 1, gsz = nout / ngrp;
        for (int i = 0; i < nc; ++i) {
            const char* res = (const char*)s->ck_pin_out + ((size_t)(c % CK_NS) * chunk + i) * hop_result_bytes(s);
            for (size_t g = 0; g < ngrp; ++g) {
                TI* dst = s->out_group > 0 ? h_out + (g * (size_t)n_hops + (size_t)(base + i)) * gsz : h_out + (size_t)(base + i) * nout;
                if (s->f64) {
                    const double* po = (const double*)res + g * gsz;
                    if (sizeof(TI) == sizeof(double)) std::memcpy(dst, po, gsz * sizeof(double));
                    else for (size_t j = 0; j < gsz; ++j) dst[j] = (TI)po[j];
                } else {
                    const float* po = (const float*)res + g * gsz;
                    if (sizeof(TI) == sizeof(float)) std::memcpy(dst, po, gsz * sizeof(float));
                    else for (size_t j = 0; j < gsz; ++j) dst[j] = (TI)po[j];
                }
            }
            const int r = scan_hop_status(h, hop_status_of(s, res), hop_first + base + i);
            if (r == APV_ERR_NOT_PD && worst != APV_ERR_NOT_PD) { worst = r; worst_msg = h->err; }
            if (r == APV_ERR_NO_CONVERGE && worst == APV_OK) { worst = r; worst_msg = h->err; }
        }
        return APV_OK;
    };
    // the spectra set of hop i of a chunk of parity par
    auto set_of = [&](int par, int i) {
        HopSpectra q;
        const size_t idx = (size_t)par * chunk + i;
        for (int p = 0; p < 4; ++p) q.X[p] = (char*)s->ck_X[p] + idx * (size_t)s->Kp * C * e2;
        for (int z = 0; z < 2; ++z) q.tspec[z] = (char*)s->ck_tspec[z] + idx * K * M * e2;
        q.inspec = (char*)s->ck_inspec + idx * 2 * K * e2;
        return q;
    };
    const size_t spec_bytes = ((size_t)s->fir_F / 2 + 1) * e2;       // one input spectrum of K1
    int last_par = 0, last_nc = 0, last_b = 0;
    std::string why;
    for (int c = 0; c < n_chunks && worst != APV_ERR_NOT_PD; ++c) {
        const int base = c * chunk, nc = std::min(chunk, n_hops - base), par = c & 1;
        double t0 = now();
        stage_in(c);                                         // staging slot c mod 3: chunk c - 3 was collected an iteration ago
        t_stage += now() - t0; t0 = now();
        const char* pin = (const char*)s->ck_pin_in + (size_t)(c % CK_NS) * chunk * 2 * H * e1;
        // ---------------- front half of the whole chunk ----------------
        // the spectra sets and linear buffers of this parity were last read by the back halves of chunk c - 2
        if (c >= 2)
            for (int b = 0; b < CK_NB; ++b) CK(hipStreamWaitEvent(s->front, s->ck_backdone[par][b], 0));
        tmark(s->front);
        CK(apv_launch_fir_chunk_spectra(f64, s->fir_F, P, H, nc, s->xhist[s->cur][0], s->xhist[s->cur][1], pin, s->xspec_chunk, s->front));
        // heads of the linear buffers: the newest N - H samples before the chunk, from the rings (first chunk) or from the tail
        // of the previous chunk's buffers
        {
            const int prev = par ^ 1, prev_nc = chunk;               // every chunk but the last is full
            auto head = [&](void* dst, const void* ring, const void* lin_prev, int rows) -> hipError_t {
                if (c == 0) return apv_launch_rows_copy(f64, rows, N - H, dst, RL, 0, RL, ring, N, (H + ring_first) % N, N, s->front);
                return apv_launch_rows_copy(f64, rows, N - H, dst, RL, 0, RL, lin_prev, RL, prev_nc * H, RL, s->front);
            };
            for (int p = 0; p < 4; ++p) CK(head(s->ck_resp[par][p], s->resp[p], s->ck_resp[prev][p], C));
            for (int z = 0; z < 2; ++z) CK(head(s->ck_tresp[par][z], s->tresp[z], s->ck_tresp[prev][z], M));
            CK(head(s->ck_in[par], s->inblk, s->ck_in[prev], 2));
        }
        {
            const void* oh[2] = {s->xhist[s->cur][0], s->xhist[s->cur][1]};
            void* nh[2] = {s->xhist[s->cur ^ 1][0], s->xhist[s->cur ^ 1][1]};
            CK(apv_launch_chunk_inputs(f64, P, H, N, nc, s->pad, RL, oh, nh, pin, s->ck_in[par], s->front));
            s->cur ^= 1;
        }
        {
            // K1: every (hop, channel) of the chunk in one launch
            const void *jh[6], *jx[6];
            void* jr[6];
            int jc[6];
            for (int p = 0; p < 4; ++p) {
                jh[p] = s->rirspec[path_zone(p)]; jx[p] = (const char*)s->xspec_chunk + spec_bytes * path_sig(p);
                jr[p] = s->ck_resp[par][p]; jc[p] = C;
            }
            for (int z = 0; z < 2; ++z) {
                jh[4 + z] = s->trirspec[z]; jx[4 + z] = (const char*)s->xspec_chunk + spec_bytes * z;
                jr[4 + z] = s->ck_tresp[par][z]; jc[4 + z] = M;
            }
            CK(apv_launch_fir_fft_chunk(f64, s->fir_F, 6, jh, jx, (long)(2 * (s->fir_F / 2 + 1)), jr, jc, P, H, RL, N - H, nc, s->front));
        }
        {
            // K2: every analysis transform of the chunk in one launch; hop i's block starts i H samples into the rows
            const HopSpectra q0 = set_of(par, 0);
            const void* jx[7];
            void* jspec[7];
            int jch[7], nj = 0;
            long jsc[7], jsk[7], jhop[7];
            for (int p = 0; p < 4; ++p) {
                const bool need = (p < 2) ? runA : runB;
                if (!need) continue;
                jx[nj] = s->ck_resp[par][p]; jspec[nj] = q0.X[p]; jch[nj] = C; jsc[nj] = s->xg; jsk[nj] = C; jhop[nj] = (long)s->Kp * C; ++nj;
            }
            for (int z = 0; z < 2; ++z) {
                jx[nj] = s->ck_tresp[par][z]; jspec[nj] = q0.tspec[z]; jch[nj] = M; jsc[nj] = 1; jsk[nj] = M; jhop[nj] = (long)K * M; ++nj;
            }
            jx[nj] = s->ck_in[par]; jspec[nj] = q0.inspec; jch[nj] = 2; jsc[nj] = K; jsk[nj] = 1; jhop[nj] = 2L * K; ++nj;
            hipError_t e = apv_launch_stft_analysis_chunk(f64, N, nj, jx, jch, jspec, jsc, jsk, RL, H, jhop, nc, s->front, &why);
            if (e != hipSuccess) return bail(APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
        }
        if (s->nch > 0) {
            for (int i = 0; i < nc; ++i) {
                const HopSpectra q = set_of(par, i);
                for (int z = 0; z < 2; ++z)
                    CK(apv_launch_perceptual_weights(f64, K, M, s->nch, q.tspec[z], s->G2, s->G2T, s->Cs, s->Ca, s->Leff, N, s->norm_mode,
                                                     s->Wgt[z], s->front));
                for (int p = 0; p < 4; ++p) {
                    const bool need = (p < 2) ? runA : runB;
                    if (need) CK(apv_launch_scale_spectra(f64, K, C, L, q.X[p], s->Wgt[path_zone(p)], s->front));
                }
                for (int z = 0; z < 2; ++z) CK(apv_launch_scale_spectra(f64, K, M, 1, q.tspec[z], s->Wgt[z], s->front));
            }
        }
        CK(hipEventRecord(s->ev_front[par], s->front));
        tmark(s->front);
        // ---------------- back halves ----------------
        // (a) the joint diagonalisations of the whole chunk as ONE launch (blockIdx.z = hop), then output spectra and synthesis hop
        //     by hop.  A hop's 2050 waves are two per SIMD, all head and tail; every input of the sixteen exists once
        //     the chunk's transforms have ended, and sixteen hops are a launch of the headline's size.  Same values per bin.
        if (batched) {
            hipStream_t b0 = bs[0];
            CK(hipStreamWaitEvent(b0, s->ev_front[par], 0));
            if (c >= 2) CK(hipStreamWaitEvent(b0, s->ck_done[(c - 2) % CK_NS], 0));        // the result slots of this parity are free
            tmark(b0);
            static const bool no_yield = getenv("APV_SIGNAL_NO_YIELD") != nullptr;       // A/B switch
            const size_t hop_w = (size_t)K * s->nV * L * wsz(h), hop_lam = (size_t)K * L * lsz(h);
            {
                GevdParams p = apv_base_params(h);
                const HopSpectra q0 = set_of(par, 0);
                const int first = runA ? 0 : 1;
                char* const obuf0 = (char*)s->ck_out + (size_t)par * chunk * hop_result_bytes(s);
                int32_t* const st0 = reinterpret_cast<int32_t*>(obuf0 + hop_out_bytes(s));
                p.x_c128 = f64;
                p.x_group = s->xg;
                p.XB = first ? q0.X[3] : q0.X[0];
                p.XD = first ? q0.X[2] : q0.X[1];
                p.d = q0.tspec[first];
                p.w = s->ck_wall[first];
                p.lam = s->ck_lamall[first];
                p.status = st0 + (size_t)first * K;
                p.n_zones = (runA && runB) ? 2 : 1;
                p.yield_issue = no_yield ? 0 : 1;
                if (p.n_zones == 2) {
                    p.XB1 = q0.X[3]; p.XD1 = q0.X[2]; p.d1 = q0.tspec[1];
                    p.w1 = s->ck_wall[1]; p.lam1 = s->ck_lamall[1]; p.status1 = st0 + K;
                }
                p.n_hops = nc;
                p.hop_X = (size_t)s->Kp * C * e2;
                p.hop_d = (size_t)K * M * e2;
                p.hop_w = hop_w;
                p.hop_lam = hop_lam;
                p.hop_status = hop_result_bytes(s);
                hipError_t e = apv_launch_gevd(p, h->cfg.compute_dtype, true, b0, &why);
                if (e != hipSuccess) return bail(APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
            }
            for (int i = 0; i < nc; ++i) {
                const size_t slot = (size_t)par * chunk + i;
                BackSchedule sch;
                sch.skip_gevd = true;
                sch.spectra_free = s->ck_k3[0];
                sch.tail_stream = s->tail;
                sch.result = (char*)s->ck_out + slot * hop_result_bytes(s);
                sch.ospec = (char*)s->ck_ospec + slot * (size_t)s->n_out * K * e2;
                sch.no_copy = true;
                void* wz[2] = {(char*)s->ck_wall[0] + i * hop_w, (char*)s->ck_wall[1] + i * hop_w};
                void* lz[2] = {(char*)s->ck_lamall[0] + i * hop_lam, (char*)s->ck_lamall[1] + i * hop_lam};
                rc = enqueue_back(h, b0, set_of(par, i), wz, lz, nullptr, sch);
                if (rc != APV_OK) return bail(rc, h->err);
                if (i + 1 == nc) CK(hipEventRecord(s->ck_backdone[par][0], b0));
                if (i + 2 >= nc && nc >= 2) tmark(b0);
                last_par = par; last_nc = nc; last_b = 0;
                s->hop++;
            }
            if (nc < 2) { tmark(b0); tmark(b0); }          // (the schedule print expects six marks per chunk)
        }
        // (b) hop by hop, alternating between the back streams (configurations the batched launch does not take)
        for (int i = 0; i < nc && !batched; ++i) {
            const int b = (int)((s->hop - hop_first) % CK_NB);
            if (i < CK_NB) {
                CK(hipStreamWaitEvent(bs[b], s->ev_front[par], 0));
                // the result slots of this parity are being copied back for chunk c - 2 (still in flight: the host collects two
                // chunks behind)
                if (c >= 2) CK(hipStreamWaitEvent(bs[b], s->ck_done[(c - 2) % CK_NS], 0));
            }
            if (i == 0) tmark(bs[b]);
            // Every hop of the two chunks in flight has result and output-spectra slots of its own, so a back stream never waits
            // for the tail stream inside a chunk: its next diagonalisation follows the previous one directly.
            const size_t slot = (size_t)par * chunk + i;
            BackSchedule sch;
            sch.spectra_free = s->ck_k3[b];                  // recorded after K3: the tail stream starts the synthesis there
            sch.tail_stream = s->tail;
            sch.result = (char*)s->ck_out + slot * hop_result_bytes(s);
            sch.ospec = (char*)s->ck_ospec + slot * (size_t)s->n_out * K * e2;
            sch.no_copy = true;
            static const bool no_yield = getenv("APV_SIGNAL_NO_YIELD") != nullptr;       // A/B switch
            sch.yield_issue = no_yield ? 0 : 1;
            sch.lspill = b > 0 ? s->ck_spill[b - 1] : nullptr;
            rc = enqueue_back(h, bs[b], set_of(par, i), wset[b], lset[b], nullptr, sch);
            if (rc != APV_OK) return bail(rc, h->err);
            if (i + CK_NB >= nc) CK(hipEventRecord(s->ck_backdone[par][b], bs[b]));   // this stream's last hop of the chunk
            if (i + 2 >= nc && nc >= 2) tmark(bs[b]);
            last_par = par; last_nc = nc; last_b = b;
            s->hop++;
        }
        // the chunk's results in ONE copy behind its last synthesis (16 hops: 6.4 MB at cfg3)
        CK(hipMemcpyAsync((char*)s->ck_pin_out + (size_t)(c % CK_NS) * chunk * hop_result_bytes(s),
                          (char*)s->ck_out + (size_t)par * chunk * hop_result_bytes(s), (size_t)nc * hop_result_bytes(s),
                          hipMemcpyDeviceToHost, s->tail));
        CK(hipEventRecord(s->ck_done[c % CK_NS], s->tail)); // every front and back half of the chunk is upstream of this copy
        tmark(s->tail);
        t_enq += now() - t0;
        // the host collects TWO chunks behind: while it waits for chunk c - 2 and copies it out, chunks c - 1 and c are queued on
        // the device, so the front half of chunk c runs beside the back halves of chunk c - 1 whatever the host is doing
        if (c > 1) {
            if ((rc = collect(c - 2)) != APV_OK) return bail(rc, h->err);
            c_done = c - 1;
        }
    }
    for (int c = c_done; c < n_chunks && (size_t)c * chunk < (size_t)(s->hop - hop_first); ++c) {
        if ((rc = collect(c)) != APV_OK) return bail(rc, h->err);
        c_done = c + 1;
    }
    // ---------------- the state n_hops per-hop calls would have left ----------------
    {
        hipError_t e = hipStreamSynchronize(s->front);
        for (int b = 1; b < CK_NB && e == hipSuccess; ++b) e = hipStreamSynchronize(bs[b]);
        if (e == hipSuccess) e = hipStreamSynchronize(s->tail);
        if (e == hipSuccess) e = hipStreamSynchronize(bs[0]);
        if (e != hipSuccess) return hipbail(e);
    }
    const long done = s->hop - hop_first;
    if (done > 0) {
        hipStream_t st = bs[0];
        // rings: the last hop's block, written at the ring offset the per-hop path (and its captured graphs) expects after
        // `done` more hops
        s->ring_off = (int)((ring_first + done * H) % N);
        const int src0 = (last_nc - 1) * H;
        for (int p = 0; p < 4; ++p) CK(apv_launch_rows_copy(f64, C, N, s->resp[p], N, s->ring_off, N, s->ck_resp[last_par][p], RL, src0, RL, st));
        for (int z = 0; z < 2; ++z) CK(apv_launch_rows_copy(f64, M, N, s->tresp[z], N, s->ring_off, N, s->ck_tresp[last_par][z], RL, src0, RL, st));
        CK(apv_launch_rows_copy(f64, 2, N, s->inblk, N, s->ring_off, N, s->ck_in[last_par], RL, src0, RL, st));
        // histories: the per-hop path alternates the two buffers every hop
        const int cur_expected = cur_first ^ (int)(done & 1);
        if (s->cur != cur_expected) {
            const size_t hb = ((size_t)P - 1 + H + s->pad) * e1;
            for (int g = 0; g < 2; ++g) CK(hipMemcpyAsync(s->xhist[cur_expected][g], s->xhist[s->cur][g], hb, hipMemcpyDeviceToDevice, st));
            s->cur = cur_expected;
        }
        // the last hop's spectra, filters and eigenvalues where the state arrays and the per-hop path keep them
        const HopSpectra q = set_of(last_par, last_nc - 1);
        for (int p = 0; p < 4; ++p) CK(hipMemcpyAsync(s->X[p], q.X[p], (size_t)s->Kp * C * e2, hipMemcpyDeviceToDevice, st));
        for (int z = 0; z < 2; ++z) CK(hipMemcpyAsync(s->tspec[z], q.tspec[z], (size_t)K * M * e2, hipMemcpyDeviceToDevice, st));
        CK(hipMemcpyAsync(s->inspec, q.inspec, (size_t)2 * K * e2, hipMemcpyDeviceToDevice, st));
        if (batched) {
            const size_t hop_w = (size_t)K * s->nV * L * wsz(h), hop_lam = (size_t)K * L * lsz(h);
            for (int z = 0; z < 2; ++z) {
                CK(hipMemcpyAsync(s->w[z], (char*)s->ck_wall[z] + (size_t)(last_nc - 1) * hop_w, hop_w, hipMemcpyDeviceToDevice, st));
                CK(hipMemcpyAsync(s->lam[z], (char*)s->ck_lamall[z] + (size_t)(last_nc - 1) * hop_lam, hop_lam, hipMemcpyDeviceToDevice, st));
            }
        } else if (last_b != 0) {
            for (int z = 0; z < 2; ++z) {
                CK(hipMemcpyAsync(s->w[z], wset[last_b][z], (size_t)K * s->nV * L * wsz(h), hipMemcpyDeviceToDevice, st));
                CK(hipMemcpyAsync(s->lam[z], lset[last_b][z], (size_t)K * L * lsz(h), hipMemcpyDeviceToDevice, st));
            }
        }
        CK(hipStreamSynchronize(st));
    }
#undef CK
    if (timing && tev.size() >= 6) {
        fprintf(stderr, "[apv signal] device schedule, ms from the first chunk's front start: chunk | front start - end | back start - end of two of its streams | copied\n");
        for (size_t c = 0; c + 1 <= tev.size() / 6; ++c) {
            float v[6];
            for (int q = 0; q < 6; ++q) (void)hipEventElapsedTime(&v[q], tev[0], tev[6 * c + q]);
            if (c < 3 || c + 3 >= tev.size() / 6)
                fprintf(stderr, "[apv signal]   %2zu | %7.3f - %7.3f | %7.3f - %7.3f, %7.3f | %7.3f\n", c, v[0], v[1], v[2], v[3], v[4], v[5]);
        }
    }
    for (hipEvent_t e : tev) (void)hipEventDestroy(e);
    if (timing)
        fprintf(stderr, "[apv signal] %d hops: host stage-in %.3f ms, enqueue %.3f ms, waiting for chunks %.3f ms, copy-out %.3f ms\n", n_hops,
                t_stage * 1e3, t_enq * 1e3, t_wait * 1e3, t_copy * 1e3);
    if (worst == APV_ERR_NOT_PD) return bail(worst, worst_msg);     // the hops behind the failing one were not run
    if (worst != APV_OK) return apv_fail(h, worst, worst_msg);      // APV_ERR_NO_CONVERGE: every hop ran, every output is written
    return APV_OK;
}

extern "C" {

int apv_stream_init(apv_handle* h, int32_t rir_len, const double* h_rir_A, const double* h_rir_B,
                    int32_t reference_index_A, int32_t reference_index_B, int32_t modeling_delay) {
    if (!h || !h_rir_A || !h_rir_B) return apv_fail(h, APV_ERR_ARG, "null argument");
    const apv_config& c = h->cfg;
    const int N = c.block_size, H = c.hop_size;
    {
        std::string why;
        if (!apv_stft_size_ok(N, &why)) return apv_fail(h, APV_ERR_ARG, why);
    }
    if (c.frontend < 0 || c.frontend > 2) return apv_fail(h, APV_ERR_ARG, "cfg.frontend must be 0 (follow compute_dtype), 1 (float32) or 2 (float64)");
    if (c.out_layout != 0 && c.out_layout != 1) return apv_fail(h, APV_ERR_ARG, "cfg.out_layout must be 0 (channel-major) or 1 (sample-major groups)");
    const int f64 = frontend_f64(c);
    if (f64 && N > 4096) return apv_fail(h, APV_ERR_ARG, "float64 front-end: block_size <= 4096 (double-precision FFT in LDS)");
    if (H < 1 || H > N) return apv_fail(h, APV_ERR_ARG, "hop_size must be in 1..block_size");
    if (c.n_bins != N / 2 + 1) return apv_fail(h, APV_ERR_ARG, "streaming handle needs n_bins == block_size/2 + 1");
    if (rir_len < 1 || modeling_delay < 0 || modeling_delay >= rir_len) return apv_fail(h, APV_ERR_ARG, "rir_len / modeling_delay out of range");
    if (reference_index_A < 0 || reference_index_A >= c.n_srcs || reference_index_B < 0 || reference_index_B >= c.n_srcs)
        return apv_fail(h, APV_ERR_ARG, "reference index out of range");
    if (c.n_zones < 1 || c.n_zones > 3) return apv_fail(h, APV_ERR_ARG, "n_zones is a bit mask: 1 = A, 2 = B, 3 = both");
    SCHK(h, hipSetDevice(h->device));
    apv_stream_free(h);
    apv_stream* s = new apv_stream();
    std::memset(static_cast<void*>(s), 0, offsetof(apv_stream, execs));
    h->st = s;
    s->N = N; s->H = H; s->K = N / 2 + 1; s->L = c.n_srcs; s->M = c.n_mics; s->C = s->L * s->M;
    s->P = rir_len; s->nV = c.n_ranks; s->zones = c.n_zones; s->pad = apv_fir_pad();
    s->f64 = f64; s->esz = f64 ? 8 : 4;
    s->ring_off = 0; s->cur = 0;
    const int nz = ((s->zones & 1) ? 1 : 0) + ((s->zones & 2) ? 1 : 0);
    s->n_out = nz * s->nV * s->L + 2 * s->L;
    s->out_group = c.out_layout == 1 ? s->L : 0;
    s->Kp = (s->K + 7) / 8 * 8;
    s->xg_default = apv_gevd_reads_groups(apv_base_params(h), h->cfg.compute_dtype, f64 != 0);
    s->xg = s->xg_default;
    const int L = s->L, M = s->M, C = s->C, P = s->P, K = s->K;
    const size_t e1 = s->esz, e2 = 2 * s->esz;
    int rc;
    // RIRs: host (P, L, M) float64 C-order -> device [P][m*L + l] in the front-end precision
    std::vector<double> tmp((size_t)P * C), ttmp((size_t)P * M);
    for (int z = 0; z < 2; ++z) {
        const double* src = z ? h_rir_B : h_rir_A;
        const int ref = z ? reference_index_B : reference_index_A;
        for (int p = 0; p < P; ++p)
            for (int l = 0; l < L; ++l)
                for (int m = 0; m < M; ++m) tmp[(size_t)p * C + m * L + l] = src[((size_t)p * L + l) * M + m];
        // target RIR: reference loudspeaker delayed by modeling_delay (apvast.py:102-112)
        std::fill(ttmp.begin(), ttmp.end(), 0.0);
        for (int p = modeling_delay; p < P; ++p)
            for (int m = 0; m < M; ++m) ttmp[(size_t)p * M + m] = src[((size_t)(p - modeling_delay) * L + ref) * M + m];
        if ((rc = dalloc(h, &s->rir[z], (size_t)P * C, e1))) return rc;
        if ((rc = dalloc(h, &s->trir[z], (size_t)P * M, e1))) return rc;
        if ((rc = upload(h, f64, s->rir[z], tmp))) return rc;
        if ((rc = upload(h, f64, s->trir[z], ttmp))) return rc;
    }
    // long responses: the hop's convolution goes through the frequency domain (fir_fft_kernel); APV_FIR_DIRECT keeps the
    // direct form on the matrix cores (A/B switch)
    s->fir_F = getenv("APV_FIR_DIRECT") == nullptr ? apv_fir_fft_size(f64, P, H) : 0;
    s->fir_np = 1;
    s->keep = P - 1;
    if (s->fir_F == 0 && getenv("APV_FIR_DIRECT") == nullptr) {
        // too long for one segment in LDS: uniformly partitioned, partitions of H taps in segments of 2 H samples
        const int np = apv_fir_partitions(f64, P, H);
        if (np > 0) {
            s->fir_F = 2 * H;
            s->fir_np = np;
            s->keep = np * H;
        }
    }
    if (s->fir_F > 0) {
        const int F = s->fir_F, np = s->fir_np;
        const size_t Kf = (size_t)F / 2 + 1;
        void* cm = nullptr;                                  // [C][P] channel-major copy of one bank, set-up only
        if ((rc = dalloc(h, &cm, (size_t)C * P, e1))) return rc;
        std::vector<double> tr((size_t)C * P);
        std::string why;
        for (int z = 0; z < 2; ++z) {
            const double* src = z ? h_rir_B : h_rir_A;
            const int ref = z ? reference_index_B : reference_index_A;
            if ((rc = dalloc(h, &s->rirspec[z], (size_t)np * Kf * C, e2))) return rc;
            if ((rc = dalloc(h, &s->trirspec[z], (size_t)np * Kf * M, e2))) return rc;
            for (int p = 0; p < P; ++p)
                for (int l = 0; l < L; ++l)
                    for (int m = 0; m < M; ++m) tr[(size_t)(m * L + l) * P + p] = src[((size_t)p * L + l) * M + m];
            if ((rc = upload(h, f64, cm, tr))) return rc;
            hipError_t e = hipSuccess;
            // (one segment: the whole response; partitioned: taps [q H, (q + 1) H) of every channel, partition-major)
            for (int q = 0; q < np && e == hipSuccess; ++q) {
                const int taps = np == 1 ? P : std::min(H, P - q * H);
                e = apv_launch_fir_spectra_part(f64, F, C, (const char*)cm + (size_t)q * (np == 1 ? 0 : H) * e1, P, taps,
                                                (char*)s->rirspec[z] + (size_t)q * Kf * C * e2, h->stream, &why);
            }
            if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
            SCHK(h, hipStreamSynchronize(h->stream));
            std::fill(tr.begin(), tr.end(), 0.0);
            for (int p = modeling_delay; p < P; ++p)
                for (int m = 0; m < M; ++m) tr[(size_t)m * P + p] = src[((size_t)(p - modeling_delay) * L + ref) * M + m];
            if ((rc = upload(h, f64, cm, tr))) return rc;
            e = hipSuccess;
            for (int q = 0; q < np && e == hipSuccess; ++q) {
                const int taps = np == 1 ? P : std::min(H, P - q * H);
                e = apv_launch_fir_spectra_part(f64, F, M, (const char*)cm + (size_t)q * (np == 1 ? 0 : H) * e1, P, taps,
                                                (char*)s->trirspec[z] + (size_t)q * Kf * M * e2, h->stream, &why);
            }
            if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
            SCHK(h, hipStreamSynchronize(h->stream));
        }
        (void)hipFree(cm);
        if ((rc = dalloc(h, &s->xspec, (size_t)np * 2 * Kf, e2))) return rc;
    }
    const size_t hist = (size_t)s->keep + H + s->pad;
    for (int b = 0; b < 2; ++b)
        for (int g = 0; g < 2; ++g)
            if ((rc = dalloc(h, &s->xhist[b][g], hist, e1))) return rc;
    for (int p = 0; p < 4; ++p) {
        if ((rc = dalloc(h, &s->resp[p], (size_t)C * N, e1))) return rc;
        if ((rc = dalloc(h, &s->X[p], (size_t)s->Kp * C, e2))) return rc;
    }
    for (int z = 0; z < 2; ++z) {
        if ((rc = dalloc(h, &s->tresp[z], (size_t)M * N, e1))) return rc;
        if ((rc = dalloc(h, &s->tspec[z], (size_t)K * M, e2))) return rc;
        if ((rc = dalloc(h, &s->w[z], (size_t)K * s->nV * L, wsz(h)))) return rc;
        if ((rc = dalloc(h, &s->lam[z], (size_t)K * L, lsz(h)))) return rc;
    }
    if ((rc = dalloc(h, &s->inblk, (size_t)2 * N, e1))) return rc;
    if ((rc = dalloc(h, &s->inspec, (size_t)2 * K, e2))) return rc;
    if ((rc = dalloc(h, &s->tgt, (size_t)L * K, e2))) return rc;
    if ((rc = dalloc(h, &s->outspec, (size_t)s->n_out * K, e2))) return rc;
    if ((rc = dalloc(h, &s->outov, (size_t)s->n_out * N, e1))) return rc;
    {
        // samples, then the status words [zone A | zone B] of the hop: one copy back
        char* outbuf = nullptr;
        if ((rc = dalloc(h, &outbuf, hop_result_bytes(s), 1))) return rc;
        s->out = outbuf;
        s->status[0] = reinterpret_cast<int32_t*>(outbuf + hop_out_bytes(s));
        s->status[1] = s->status[0] + K;
    }
    // target filter spectra: rfft of a unit impulse at tap modeling_delay of the A reference loudspeaker
    // (apvast.py:389-390, 418, 422: the same filter serves A_t and B_t)
    std::vector<double> tg((size_t)L * K * 2, 0.0);
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < K; ++k) {
        // exact at the multiples of a quarter turn, like an FFT of the unit impulse
        const long q = ((long)k * modeling_delay) % N;
        double cr = std::cos(-2.0 * PI * (double)q / (double)N), ci = std::sin(-2.0 * PI * (double)q / (double)N);
        if ((4 * q) % N == 0) {
            const int quarter = (int)((4 * q) / N);
            cr = quarter == 0 ? 1.0 : quarter == 2 ? -1.0 : 0.0;
            ci = quarter == 1 ? -1.0 : quarter == 3 ? 1.0 : 0.0;
        }
        tg[((size_t)reference_index_A * K + k) * 2] = cr;
        tg[((size_t)reference_index_A * K + k) * 2 + 1] = ci;
    }
    if ((rc = upload(h, f64, s->tgt, tg))) return rc;
    s->h_status.assign((size_t)2 * K, 0);
    SCHK(h, hipHostMalloc((void**)&s->pin_in, e1 * 2 * H, hipHostMallocDefault));
    SCHK(h, hipHostMalloc((void**)&s->pin_out, hop_result_bytes(s), hipHostMallocDefault));
    std::memset(s->pin_out, 0, hop_result_bytes(s));
    // the launch sequence of a hop depends on (ring_off, cur) only: ring_off has period N / gcd(N, H), cur period 2
    {
        int a = N, b = H;
        while (b) { const int t = a % b; a = b; b = t; }
        int per = N / a;
        if (per % 2) per *= 2;
        s->period = (per <= 16 && getenv("APV_NO_GRAPH") == nullptr) ? per : 0;
        s->execs.assign(s->period > 0 ? s->period : 0, nullptr);
    }
    s->hop = 0;
    SCHK(h, apv_stft_prepare(N, f64));
    SCHK(h, hipStreamSynchronize(h->stream));
    return APV_OK;
}

int apv_process_block(apv_handle* h, const float* h_in_A, const float* h_in_B, float* h_out) {
    return process_block_t<float>(h, h_in_A, h_in_B, h_out);
}

int apv_process_block_f64(apv_handle* h, const double* h_in_A, const double* h_in_B, double* h_out) {
    return process_block_t<double>(h, h_in_A, h_in_B, h_out);
}

int apv_process_signal(apv_handle* h, int32_t n_hops, const float* h_in_A, const float* h_in_B, float* h_out) {
    return process_signal_t<float>(h, n_hops, h_in_A, h_in_B, h_out);
}

int apv_process_signal_f64(apv_handle* h, int32_t n_hops, const double* h_in_A, const double* h_in_B, double* h_out) {
    return process_signal_t<double>(h, n_hops, h_in_A, h_in_B, h_out);
}

int apv_stream_is_f64(apv_handle* h) { return (h && h->st) ? h->st->f64 : -1; }

long apv_stream_not_converged(apv_handle* h) {
    if (h && h->st) return h->st->not_converged;
    if (h && h->bb) return apv_bb_not_converged(h);
    return -1;
}

// Per-bin statistics and eigenvectors of the CURRENT hop, recomputed in float64 on demand from the hop's control-point
// spectra (nothing on the per-hop path pays for them).  zone 0: bright A->A, dark A->B, target A; zone 1: B->B, B->A, B.
int apv_stream_get_statistics(apv_handle* h, int32_t zone, double* h_RB, double* h_RD, double* h_r, double* h_U,
                              double* h_lam) {
    if (!h || !h->st) return apv_fail(h, APV_ERR_ARG, "apv_stream_init has not been called");
    if (zone != 0 && zone != 1) return apv_fail(h, APV_ERR_ARG, "zone must be 0 (A) or 1 (B)");
    apv_stream* s = h->st;
    if (!(s->zones & (1 << zone))) return apv_fail(h, APV_ERR_STATE, "this zone program does not run (run_A / run_B)");
    SCHK(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int K = s->K, L = s->L, M = s->M;
    const size_t mat = (size_t)K * L * L * 16, vec = (size_t)K * L * 16;
    // the work space is sized by (K, L) alone, which a stream never changes: allocated once, kept until apv_stream_free
    void*& dRB = s->stat_ws[0]; void*& dRD = s->stat_ws[1]; void*& dr = s->stat_ws[2]; void*& dU = s->stat_ws[3];
    void*& dw = s->stat_ws[4]; void*& dl = s->stat_ws[5]; void*& dspill = s->stat_ws[6]; void*& dst = s->stat_ws[7];
    if (!dRB) SCHK(h, hipMalloc(&dRB, mat));
    if (!dRD) SCHK(h, hipMalloc(&dRD, mat));
    if (!dr) SCHK(h, hipMalloc(&dr, vec));
    const void* XB = zone ? s->X[3] : s->X[0];
    const void* XD = zone ? s->X[2] : s->X[1];
    if (s->xg > 1) {
        // grouped spectra: the correlation kernel reads bin-major slabs, so the two sets are regrouped into scratch first
        void*& dgb = s->stat_ws[8]; void*& dgd = s->stat_ws[9];
        const size_t sb = (size_t)K * s->C * 2 * s->esz;
        if (!dgb) SCHK(h, hipMalloc(&dgb, sb));
        if (!dgd) SCHK(h, hipMalloc(&dgd, sb));
        SCHK(h, apv_launch_ungroup_spectra(s->f64, K, s->C, s->xg, XB, dgb, st));
        SCHK(h, apv_launch_ungroup_spectra(s->f64, K, s->C, s->xg, XD, dgd, st));
        XB = dgb;
        XD = dgd;
    }
    hipError_t e = s->f64 ? apv_launch_corr_c128(K, M, L, (const double2*)XB, (const double2*)XD, (const double2*)s->tspec[zone],
                                                 (double2*)dRB, (double2*)dRD, (double2*)dr, st)
                          : apv_launch_corr(APV_F64, K, M, L, (const float2*)XB, (const float2*)XD, (const float2*)s->tspec[zone], dRB, dRD,
                                            dr, st);
    if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, std::string("statistics: ") + hipGetErrorString(e));
    if (h_U || h_lam) {
        if (!dU) SCHK(h, hipMalloc(&dU, mat));
        if (!dw) SCHK(h, hipMalloc(&dw, vec));
        if (!dl) SCHK(h, hipMalloc(&dl, (size_t)K * L * 8));
        if (!dst) SCHK(h, hipMalloc(&dst, (size_t)K * 4));
        GevdParams p = apv_base_params(h);
        const size_t sb = apv_gevd_spill_bytes(L, K, APV_F64, p.reg_mode, p.reg_bright, p.sweep_tol2, 1);
        if (sb > s->stat_spill_bytes) {
            if (dspill) SCHK(h, hipFree(dspill));
            dspill = nullptr;
            s->stat_spill_bytes = 0;
            SCHK(h, hipMalloc(&dspill, sb));
            s->stat_spill_bytes = sb;
        }
        p.nV = 1; p.ranks[0] = 1; p.out_c128 = 1; p.n_zones = 1;
        p.RB = dRB; p.RD = dRD; p.r = dr; p.w = dw; p.lam = dl; p.status = (int32_t*)dst; p.U = dU; p.Lspill = dspill;
        std::string why;
        e = apv_launch_gevd(p, APV_F64, false, st, &why);
        if (e != hipSuccess) return apv_fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
    }
    if (h_RB) SCHK(h, hipMemcpyAsync(h_RB, dRB, mat, hipMemcpyDeviceToHost, st));
    if (h_RD) SCHK(h, hipMemcpyAsync(h_RD, dRD, mat, hipMemcpyDeviceToHost, st));
    if (h_r) SCHK(h, hipMemcpyAsync(h_r, dr, vec, hipMemcpyDeviceToHost, st));
    if (h_U) SCHK(h, hipMemcpyAsync(h_U, dU, mat, hipMemcpyDeviceToHost, st));
    if (h_lam) SCHK(h, hipMemcpyAsync(h_lam, dl, (size_t)K * L * 8, hipMemcpyDeviceToHost, st));
    SCHK(h, hipStreamSynchronize(st));
    return APV_OK;
}

// Enable (n_channels > 0) or disable (0) the perceptual weighting.  h_G2 [K][n_channels]: squared
// outer/middle-ear x gammatone responses (perceptualModel.m:52-54); Cs, Ca, Leff: perceptualModel.m:57, 114-115;
// normalisation 0: unit vector over the K bins (apvast.py:322-324), 1: over the full symmetric curve
// (perceptualModel.m:177-190).                                        replaces apvast.py:313-324 / apVast.m:386-408
int apv_stream_set_perceptual(apv_handle* h, int32_t n_channels, const double* h_G2, double Cs, double Ca, double Leff,
                              int32_t normalisation) {
    if (!h || !h->st) return apv_fail(h, APV_ERR_ARG, "apv_stream_init has not been called");
    apv_stream* s = h->st;
    SCHK(h, hipSetDevice(h->device));
    SCHK(h, hipStreamSynchronize(h->stream));
    // the captured launch sequences depend on whether the weighting is on
    for (hipGraphExec_t& e : s->execs) {
        if (e) (void)hipGraphExecDestroy(e);
        e = nullptr;
    }
    if (n_channels <= 0) {
        s->nch = 0;
        s->xg = s->xg_default;
        return APV_OK;
    }
    s->xg = 1;                   // the weighting's kernels scale bin-major spectra
    if (!h_G2 || n_channels > 512 || (normalisation != 0 && normalisation != 1)) return apv_fail(h, APV_ERR_ARG, "bad perceptual tables");
    const int K = s->K;
    void* old[] = {s->G2, s->G2T, s->Wgt[0], s->Wgt[1]};
    for (void* b : old)
        if (b) (void)hipFree(b);
    s->G2 = s->G2T = nullptr;
    s->Wgt[0] = s->Wgt[1] = nullptr;
    std::vector<double> gt((size_t)n_channels * K);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < n_channels; ++i) gt[(size_t)i * K + k] = h_G2[(size_t)k * n_channels + i];
    SCHK(h, hipMalloc((void**)&s->G2, sizeof(double) * (size_t)K * n_channels));
    SCHK(h, hipMalloc((void**)&s->G2T, sizeof(double) * (size_t)K * n_channels));
    for (int z = 0; z < 2; ++z) SCHK(h, hipMalloc((void**)&s->Wgt[z], s->esz * (size_t)K * s->M));
    // on the handle's stream, like every other upload of this file (a null-stream copy is not ordered with it)
    SCHK(h, hipMemcpyAsync(s->G2, h_G2, sizeof(double) * (size_t)K * n_channels, hipMemcpyHostToDevice, h->stream));
    SCHK(h, hipMemcpyAsync(s->G2T, gt.data(), sizeof(double) * gt.size(), hipMemcpyHostToDevice, h->stream));
    SCHK(h, hipStreamSynchronize(h->stream));               // gt goes out of scope; h_G2 is the caller's
    s->nch = n_channels; s->Cs = Cs; s->Ca = Ca; s->Leff = Leff; s->norm_mode = normalisation;
    return APV_OK;
}

// Named state arrays, as stored on the device: real samples are float32 (float64 with the float64 front-end),
// spectra the matching complex type; rings are returned in LOGICAL order:
//   "response<p>"     [C][N]           "target_response<z>" [M][N]       "input_block" [2][N]
//   "input_history<g>" [keep+H] (keep = P-1; fir_np H when K1 is partitioned: apv_state_bytes tells)   "out_overlap" [n_out][N]
//   "spectra<p>"      [K][C] complex   "target_spectra<z>"  [K][M] complex   "input_spectrum" [2][K] complex
//   "weights<z>"      [K][M]
//   "w_A" / "w_B"     [K][nV][L] c64|c128        "lambda_A" / "lambda_B" [K][L] f32|f64   (cfg.out_c128)
static int state_lookup(apv_handle* h, const char* name, void** dptr, size_t* bytes, int* ring_rows) {
    apv_stream* s = h->st;
    *ring_rows = 0;
    const std::string n(name);
    const size_t N = s->N, K = s->K, C = s->C, M = s->M, L = s->L, e1 = s->esz, e2 = 2 * s->esz;
    if (n.rfind("response", 0) == 0 && n.size() == 9 && n[8] >= '0' && n[8] <= '3') {
        *dptr = s->resp[n[8] - '0']; *bytes = C * N * e1; *ring_rows = (int)C; return APV_OK; }
    if (n == "target_response0" || n == "target_response1") {
        *dptr = s->tresp[n.back() - '0']; *bytes = M * N * e1; *ring_rows = (int)M; return APV_OK; }
    if (n == "input_block") { *dptr = s->inblk; *bytes = 2 * N * e1; *ring_rows = 2; return APV_OK; }
    if (n == "input_history0" || n == "input_history1") {
        *dptr = s->xhist[s->cur][n.back() - '0']; *bytes = (size_t)(s->keep + s->H) * e1; return APV_OK; }
    if (n == "out_overlap") { *dptr = s->outov; *bytes = (size_t)s->n_out * N * e1; return APV_OK; }
    if (n.rfind("spectra", 0) == 0 && n.size() == 8 && n[7] >= '0' && n[7] <= '3') {
        *dptr = s->X[n[7] - '0']; *bytes = K * C * e2; *ring_rows = s->xg > 1 ? -1 : 0; return APV_OK; }      // -1: grouped on the device
    if (n == "target_spectra0" || n == "target_spectra1") { *dptr = s->tspec[n.back() - '0']; *bytes = K * M * e2; return APV_OK; }
    if (n == "input_spectrum") { *dptr = s->inspec; *bytes = 2 * K * e2; return APV_OK; }
    if ((n == "weights0" || n == "weights1") && s->nch > 0) { *dptr = s->Wgt[n.back() - '0']; *bytes = K * M * e1; return APV_OK; }
    if (n == "w_A" || n == "w_B") { *dptr = s->w[n == "w_B"]; *bytes = K * s->nV * L * wsz(h); return APV_OK; }
    if (n == "lambda_A" || n == "lambda_B") { *dptr = s->lam[n == "lambda_B"]; *bytes = K * L * lsz(h); return APV_OK; }
    return apv_fail(h, APV_ERR_STATE, std::string("unknown state name: ") + name);
}

int apv_state_bytes(apv_handle* h, const char* name, size_t* bytes) {
    if (!h || !h->st || !name || !bytes) return apv_fail(h, APV_ERR_ARG, "null argument / no stream");
    void* d; int rr;
    return state_lookup(h, name, &d, bytes, &rr);
}

int apv_get_state(apv_handle* h, const char* name, void* h_dst, size_t bytes) {
    if (!h || !h->st || !name || !h_dst) return apv_fail(h, APV_ERR_ARG, "null argument / no stream");
    void* d; size_t need; int rr;
    int rc = state_lookup(h, name, &d, &need, &rr);
    if (rc != APV_OK) return rc;
    if (bytes != need) return apv_fail(h, APV_ERR_STATE, "state size mismatch");
    SCHK(h, hipSetDevice(h->device));
    if (rr == 0) {
        SCHK(h, hipMemcpyAsync(h_dst, d, need, hipMemcpyDeviceToHost, h->stream));
        SCHK(h, hipStreamSynchronize(h->stream));
        return APV_OK;
    }
    if (rr < 0) {
        // control-point spectra in the grouped layout [Kp / g][C][g]: the caller gets them bin-major [K][C], as documented
        const apv_stream* s = h->st;
        const size_t K = s->K, C = s->C, g = s->xg, e2 = 2 * s->esz;
        std::vector<char> tmp((size_t)s->Kp * C * e2);
        SCHK(h, hipMemcpyAsync(tmp.data(), d, tmp.size(), hipMemcpyDeviceToHost, h->stream));
        SCHK(h, hipStreamSynchronize(h->stream));
        char* out = (char*)h_dst;
        for (size_t k = 0; k < K; ++k)
            for (size_t c = 0; c < C; ++c)
                std::memcpy(out + (k * C + c) * e2, tmp.data() + ((k / g) * g * C + c * g + k % g) * e2, e2);
        return APV_OK;
    }
    // ring: rotate rows into logical order
    const size_t N = h->st->N, off = h->st->ring_off, e1 = h->st->esz;
    std::vector<char> tmp(need);
    SCHK(h, hipMemcpyAsync(tmp.data(), d, need, hipMemcpyDeviceToHost, h->stream));
    SCHK(h, hipStreamSynchronize(h->stream));
    char* out = (char*)h_dst;
    for (int r = 0; r < rr; ++r) {
        // logical [0, N - off) = physical [off, N); logical [N - off, N) = physical [0, off)
        std::memcpy(out + ((size_t)r * N) * e1, tmp.data() + ((size_t)r * N + off) * e1, (N - off) * e1);
        std::memcpy(out + ((size_t)r * N + (N - off)) * e1, tmp.data() + ((size_t)r * N) * e1, off * e1);
    }
    return APV_OK;
}

int apv_set_state(apv_handle* h, const char* name, const void* h_src, size_t bytes) {
    if (!h || !h->st || !name || !h_src) return apv_fail(h, APV_ERR_ARG, "null argument / no stream");
    void* d; size_t need; int rr;
    int rc = state_lookup(h, name, &d, &need, &rr);
    if (rc != APV_OK) return rc;
    if (bytes != need) return apv_fail(h, APV_ERR_STATE, "state size mismatch");
    SCHK(h, hipSetDevice(h->device));
    if (rr < 0) return apv_fail(h, APV_ERR_STATE, "the control-point spectra are recomputed by every hop and cannot be set");
    if (rr == 0) {
        SCHK(h, hipMemcpyAsync(d, h_src, need, hipMemcpyHostToDevice, h->stream));
        SCHK(h, hipStreamSynchronize(h->stream));
        return APV_OK;
    }
    const size_t N = h->st->N, off = h->st->ring_off, e1 = h->st->esz;
    std::vector<char> tmp(need);
    const char* in = (const char*)h_src;
    for (int r = 0; r < rr; ++r) {
        std::memcpy(tmp.data() + ((size_t)r * N + off) * e1, in + ((size_t)r * N) * e1, (N - off) * e1);
        std::memcpy(tmp.data() + ((size_t)r * N) * e1, in + ((size_t)r * N + (N - off)) * e1, off * e1);
    }
    SCHK(h, hipMemcpyAsync(d, tmp.data(), need, hipMemcpyHostToDevice, h->stream));
    SCHK(h, hipStreamSynchronize(h->stream));               // tmp goes out of scope
    return APV_OK;
}

}  // extern "C"
