// C ABI of libapvast_hip.so (see include/apvast_hip.h for the contract).
#include "apv_internal.h"
#include <utility>

#include <cstdlib>

#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>

namespace {

thread_local std::string g_create_err;

int fail(apv_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

int hipfail(apv_handle* h, hipError_t e, const char* what) {
    return fail(h, APV_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIPCHK(h, call)                                   \
    do {                                                  \
        hipError_t _e = (call);                           \
        if (_e != hipSuccess) return hipfail(h, _e, #call); \
    } while (0)

size_t csize(const apv_handle* h) { return h->cfg.compute_dtype == APV_F64 ? 16 : 8; }
size_t wsize(const apv_handle* h) { return h->cfg.out_c128 ? 16 : 8; }
size_t lsize(const apv_handle* h) { return h->cfg.out_c128 ? 8 : 4; }

}  // namespace

int apv_fail(apv_handle* h, int code, const std::string& msg) { return fail(h, code, msg); }

GevdParams apv_base_params(const apv_handle* h);

namespace {

GevdParams base_params(const apv_handle* h) { return apv_base_params(h); }

}  // namespace

GevdParams apv_base_params(const apv_handle* h) {
    GevdParams p;
    std::memset(&p, 0, sizeof(p));
    const apv_config& c = h->cfg;
    p.n = c.n_srcs;
    p.M = c.n_mics;
    p.K = c.n_bins;
    p.nV = c.n_ranks;
    for (int i = 0; i < c.n_ranks; ++i) p.ranks[i] = c.ranks[i];
    p.mu = c.mu;
    p.reg_dark = c.reg_dark;
    p.reg_bright = c.reg_bright;
    p.reg_mode = c.reg_mode;
    p.max_sweeps = c.max_sweeps;
    p.debug_stop = c.debug_stop;
    p.sweep_tol2 = c.sweep_tol2;
    p.out_c128 = c.out_c128;
    p.Lspill = h->d_Lspill;
    p.stamps = h->d_stamps;
    return p;
}

namespace {

int ensure_spill(apv_handle* h, int n, int K) {
    const apv_config& c = h->cfg;
    const size_t need = apv_gevd_spill_bytes(n, K, c.compute_dtype, c.reg_mode, c.reg_bright, c.sweep_tol2, c.n_zones == 3 ? 2 : 1);
    if (need > h->lspill_bytes) {
        if (h->d_Lspill) HIPCHK(h, hipFree(h->d_Lspill));
        h->d_Lspill = nullptr;
        h->lspill_bytes = 0;
        if (h->d_Lspill_lane1) {          // (too small now: launches that need the scratch keep to one lane until pipelining is set again)
            HIPCHK(h, hipFree(h->d_Lspill_lane1));
            h->d_Lspill_lane1 = nullptr;
        }
        HIPCHK(h, hipMalloc(&h->d_Lspill, need));
        h->lspill_bytes = need;
    }
    return APV_OK;
}

// scratch R of the split float32 update at orders 32 / 64 (apv_update_dev): correlation on the f32 matrix cores -> [2][K][L][L] +
// [K][L] c64 -> LDS kernel.  Sized at apv_create so that the per-block path never allocates.
bool split_f32_update(const apv_config& c) {
    const bool split64 = c.n_srcs == 64 && !apv_gevd64_eligible(c.n_srcs, c.reg_mode, c.reg_bright, c.sweep_tol2);
    return c.compute_dtype == APV_F32 && (c.n_srcs == 32 || split64) && (c.n_mics % 2) == 0 && c.n_mics >= 8;
}

int ensure_rscratch(apv_handle* h) {
    const apv_config& c = h->cfg;
    if (!split_f32_update(c)) return APV_OK;
    const size_t K = c.n_bins, L = c.n_srcs;
    const size_t need = (2 * K * L * L + K * L) * 8;
    if (need > h->rscratch_bytes) {
        if (h->d_Rscratch) HIPCHK(h, hipFree(h->d_Rscratch));
        h->d_Rscratch = nullptr;
        h->rscratch_bytes = 0;
        HIPCHK(h, hipMalloc(&h->d_Rscratch, need ? need : 1));
        h->rscratch_bytes = need;
    }
    return APV_OK;
}

int ensure_staging(apv_handle* h) {
    if (h->d_XB) return APV_OK;
    const apv_config& c = h->cfg;
    const size_t K = c.n_bins, M = c.n_mics, L = c.n_srcs;
    HIPCHK(h, hipMalloc(&h->d_XB, K * M * L * 8));
    HIPCHK(h, hipMalloc(&h->d_XD, K * M * L * 8));
    HIPCHK(h, hipMalloc(&h->d_d, K * M * 8));
    HIPCHK(h, hipMalloc(&h->d_w, K * c.n_ranks * L * 16));
    HIPCHK(h, hipMalloc(&h->d_lam, K * L * 8));
    HIPCHK(h, hipMalloc((void**)&h->d_status, K * sizeof(int32_t)));
    return APV_OK;
}


// ---- update lanes (apv_set_update_streams) -----------------------------------------------------------------------------------
// With two lanes (the handle's stream and one more), launch i + 1 of apv_update_dev starts on the other lane while the waves of launch i's last round are still
// finishing (a cfg2 launch is 8 rounds of 4 waves per SIMD; 3.6 of 4 alive on average: profiles/r03/stage_stamps_32768_b.md).
// Ordering is by events, never by the host:
//   lanes_join   the control stream waits for the latest launch of every lane (before anything it is asked to do with buffers)
//   lanes_fork   a lane waits for what the control stream has been given since the lanes last looked (ctrl_dirty)
//   operand ranges of the latest launch of the OTHER lane: a launch that writes what it reads or writes, or reads what it writes, waits
//   for it; and every launch waits for the other lane's launch BEFORE its latest, so that the latest is the only launch of the other
//   lane it can ever run beside (the ranges of one launch per lane are then all there is to compare; in the steady state that
//   event completed a launch ago)
bool lanes_on(const apv_handle* h) { return h->n_lanes > 1; }

int lanes_join(apv_handle* h, bool dirties) {
    if (!lanes_on(h)) return APV_OK;
    for (auto& ln : h->lane)
        if (ln.used) HIPCHK(h, hipStreamWaitEvent(h->stream, ln.ev, 0));
    if (dirties) h->ctrl_dirty = true;
    return APV_OK;
}

bool ranges_meet(const void* a, size_t na, const void* b, size_t nb) {
    if (!a || !b || na == 0 || nb == 0) return false;
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + nb && b0 < a0 + na;
}

int lanes_sync(apv_handle* h) {
    for (auto& ln : h->lane)
        if (ln.s) HIPCHK(h, hipStreamSynchronize(ln.s));
    return APV_OK;
}

}  // namespace

// The blocks apv_host_alloc has handed out (page-locked, visible to every device): a result pointer inside one of them takes a
// device-to-host copy by DMA.  A table of our own, because asking the runtime about an arbitrary pointer
// (hipPointerGetAttributes) costs a search and, for pageable memory, an error path on every call.
namespace {
std::mutex g_host_blocks_mu;
std::map<uintptr_t, size_t> g_host_blocks;
}  // namespace

void apv_host_blocks_note(const void* p, size_t bytes, bool add) {
    std::lock_guard<std::mutex> lk(g_host_blocks_mu);
    if (add) g_host_blocks[(uintptr_t)p] = bytes;
    else g_host_blocks.erase((uintptr_t)p);
}

bool apv_host_block_contains(const void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_host_blocks_mu);
    auto it = g_host_blocks.upper_bound((uintptr_t)p);
    if (it == g_host_blocks.begin()) return false;
    --it;
    return (uintptr_t)p >= it->first && (uintptr_t)p + bytes <= it->first + it->second;
}

extern "C" {

int apv_abi_version(void) { return APV_ABI_VERSION; }

const char* apv_last_error(const apv_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int apv_create(const apv_config* cfg, apv_handle** out) {
    if (!cfg || !out) return fail(nullptr, APV_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->abi_version != APV_ABI_VERSION) return fail(nullptr, APV_ERR_ARG, "ABI version mismatch");
    if (cfg->n_srcs < 1 || cfg->n_srcs > APV_MAX_N) return fail(nullptr, APV_ERR_ARG, "n_srcs must be in 1..64");
    if (cfg->n_bins < 0 || cfg->n_mics < 1) return fail(nullptr, APV_ERR_ARG, "n_bins/n_mics out of range");
    if (cfg->n_ranks < 1 || cfg->n_ranks > APV_MAX_RANKS) return fail(nullptr, APV_ERR_ARG, "n_ranks must be in 1..64");
    for (int i = 0; i < cfg->n_ranks; ++i) {
        if (cfg->ranks[i] < 1 || cfg->ranks[i] > cfg->n_srcs) return fail(nullptr, APV_ERR_ARG, "rank V out of 1..L");
        if (i && cfg->ranks[i] <= cfg->ranks[i - 1]) return fail(nullptr, APV_ERR_ARG, "ranks must be ascending");
    }
    if (cfg->compute_dtype != APV_F32 && cfg->compute_dtype != APV_F64)
        return fail(nullptr, APV_ERR_ARG, "compute_dtype must be APV_F32 or APV_F64");
    if (cfg->reg_mode != APV_REG_ABS && cfg->reg_mode != APV_REG_REL)
        return fail(nullptr, APV_ERR_ARG, "reg_mode must be APV_REG_ABS or APV_REG_REL");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(nullptr, APV_ERR_HIP, "no HIP device visible");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, APV_ERR_ARG, "device ordinal out of range");
    apv_handle* h = new (std::nothrow) apv_handle();
    if (!h) return fail(nullptr, APV_ERR_HIP, "out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    h->stream = nullptr;
    h->ev0 = h->ev1 = nullptr;
    h->d_XB = h->d_XD = h->d_d = h->d_w = h->d_lam = nullptr;
    h->d_status = nullptr;
    h->d_Lspill = nullptr;
    h->lspill_bytes = 0;
    h->d_Rscratch = nullptr;
    h->rscratch_bytes = 0;
    h->d_stamps = nullptr;
    h->st = nullptr;
    h->bb = nullptr;
    h->gl_ws = nullptr;
    h->gl_tol2 = 0.0;
    h->gl_lead_rank = 0;
    h->gl_lead_done = 0;
    h->lead_ws = nullptr;
    h->comm_stream = nullptr;
    h->ev_ready = nullptr;
    for (auto& g : h->gather_done) { g.ptr = nullptr; g.ev = nullptr; }
    h->gather_next = 0;
    h->ev_ag0 = h->ev_ag1 = nullptr;
    h->ag_bytes = 0;
    h->d_bar = nullptr;
    h->comm = nullptr;
    h->comm_rank = 0;
    h->comm_world = 1;
    for (auto& ln : h->lane) {
        ln.s = nullptr;
        ln.ev = ln.ev_prev = nullptr;
        ln.used = ln.used_prev = ln.need_fork = false;
        for (int i = 0; i < 3; ++i) { ln.rd[i] = ln.wr[i] = nullptr; ln.rd_bytes[i] = ln.wr_bytes[i] = 0; }
    }
    h->d_Lspill_lane1 = nullptr;
    h->n_lanes = 1;
    h->lane_next = 0;
    h->ctrl_dirty = false;
    h->ev_fork = nullptr;
#define CR(call)                                                         \
    do {                                                                 \
        hipError_t _e = (call);                                          \
        if (_e != hipSuccess) {                                          \
            g_create_err = std::string(#call) + ": " + hipGetErrorString(_e); \
            delete h;                                                    \
            return APV_ERR_HIP;                                          \
        }                                                                \
    } while (0)
    CR(hipSetDevice(h->device));
    CR(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CR(hipEventCreate(&h->ev0));
    CR(hipEventCreate(&h->ev1));
#undef CR
    int rc = ensure_spill(h, cfg->n_srcs, cfg->n_bins);
    if (rc == APV_OK) rc = ensure_rscratch(h);
    if (rc != APV_OK) {
        g_create_err = h->err;
        delete h;
        return rc;
    }
    *out = h;
    return APV_OK;
}

int apv_destroy(apv_handle* h) {
    if (!h) return APV_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    apv_stream_free(h);
    apv_bb_free(h);
    apv_gevd_large_free(h);
    apv_gevd_lead_free(h);
    if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
    for (auto& ln : h->lane) {
        if (ln.s) (void)hipStreamSynchronize(ln.s);
        if (ln.ev) (void)hipEventDestroy(ln.ev);
        if (ln.ev_prev) (void)hipEventDestroy(ln.ev_prev);
        if (ln.s && ln.s != h->stream) (void)hipStreamDestroy(ln.s);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->comm) ncclCommDestroy((ncclComm_t)h->comm);
    for (auto& g : h->gather_done)
        if (g.ev) (void)hipEventDestroy(g.ev);
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    if (h->ev_ag0) (void)hipEventDestroy(h->ev_ag0);
    if (h->ev_ag1) (void)hipEventDestroy(h->ev_ag1);
    if (h->d_bar) (void)hipFree(h->d_bar);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    void* bufs[] = {h->d_XB, h->d_XD, h->d_d, h->d_w, h->d_lam, h->d_status, h->d_Lspill, h->d_Lspill_lane1, h->d_Rscratch};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return APV_OK;
}

int apv_dev_alloc(apv_handle* h, size_t bytes, void** d_ptr) {
    if (!h || !d_ptr) return APV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMalloc(d_ptr, bytes ? bytes : 1));
    return APV_OK;
}

int apv_dev_free(apv_handle* h, void* d_ptr) {
    if (!h) return APV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipFree(d_ptr));
    return APV_OK;
}

int apv_memcpy_h2d(apv_handle* h, void* d_dst, const void* h_src, size_t bytes) {
    if (!h) return APV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;          // a launch in flight may still read what this copy overwrites
    HIPCHK(h, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, h->stream));
    return APV_OK;
}

int apv_memcpy_d2h(apv_handle* h, void* h_dst, const void* d_src, size_t bytes) {
    if (!h) return APV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;          // (dirty: a later launch must not overwrite the source under the copy)
    HIPCHK(h, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    return APV_OK;
}

int apv_sync(apv_handle* h) {
    if (!h) return APV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_sync(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->comm_stream) HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    return APV_OK;
}

int apv_timer_start(apv_handle* h) {
    if (!h) return APV_ERR_ARG;
    if (int rc = lanes_join(h, false)) return rc;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    return APV_OK;
}

int apv_timer_stop(apv_handle* h, float* elapsed_ms) {
    if (!h || !elapsed_ms) return APV_ERR_ARG;
    if (int rc = lanes_join(h, false)) return rc;          // the interval ends behind the latest launch of every lane
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return APV_OK;
}

int apv_update_dev(apv_handle* h, const void* d_XB, const void* d_XD, const void* d_d, void* d_w, void* d_lam,
                   int32_t* d_status) {
    if (!h) return APV_ERR_ARG;
    if (h->cfg.n_bins == 0) return APV_OK;
    if (!d_XB || !d_XD || !d_d || !d_w) return fail(h, APV_ERR_ARG, "null device pointer");
    HIPCHK(h, hipSetDevice(h->device));
    const apv_config& c = h->cfg;
    // the stream of this launch: the handle's own, or (apv_set_update_streams) the next lane.  Orders 33..64 park per-bin state in
    // scratch slots: lane 1 has slots of its own (d_Lspill_lane1).  The split float32 update shares ONE scratch R: its launches
    // all take lane 0, one after the other.
    hipStream_t st = h->stream;
    apv_handle::UpdateLane* ln = nullptr;
    if (lanes_on(h)) {
        const bool shares_scratch = (h->d_Lspill != nullptr && h->d_Lspill_lane1 == nullptr) || split_f32_update(c);
        const int li = shares_scratch ? 0 : h->lane_next;
        if (!shares_scratch) h->lane_next ^= 1;
        ln = &h->lane[li];
        st = ln->s;
        if (h->ctrl_dirty) {              // copies (or other work) went to the control stream since the lanes last looked
            HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
            for (auto& l2 : h->lane) l2.need_fork = true;
            h->ctrl_dirty = false;
        }
        if (ln->need_fork) {
            HIPCHK(h, hipStreamWaitEvent(st, h->ev_fork, 0));
            ln->need_fork = false;
        }
        const size_t K = c.n_bins, M = c.n_mics, L = c.n_srcs;
        const void* rd[3] = {d_XB, d_XD, d_d};
        const size_t rd_bytes[3] = {K * M * L * 8, K * M * L * 8, K * M * 8};
        const void* wr[3] = {d_w, d_lam, d_status};
        const size_t wr_bytes[3] = {K * c.n_ranks * L * wsize(h), d_lam ? K * L * lsize(h) : 0, d_status ? K * sizeof(int32_t) : 0};
        apv_handle::UpdateLane& other = h->lane[li ^ 1];
        if (other.used_prev) HIPCHK(h, hipStreamWaitEvent(st, other.ev_prev, 0));
        if (other.used) {
            bool clash = false;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    clash = clash || ranges_meet(wr[i], wr_bytes[i], other.wr[j], other.wr_bytes[j]) ||
                            ranges_meet(wr[i], wr_bytes[i], other.rd[j], other.rd_bytes[j]) ||
                            ranges_meet(rd[i], rd_bytes[i], other.wr[j], other.wr_bytes[j]);
            if (clash) HIPCHK(h, hipStreamWaitEvent(st, other.ev, 0));
        }
        for (int i = 0; i < 3; ++i) {
            ln->rd[i] = rd[i]; ln->rd_bytes[i] = rd_bytes[i];
            ln->wr[i] = wr[i]; ln->wr_bytes[i] = wr_bytes[i];
        }
    }
    for (auto& g : h->gather_done)
        if (g.ptr == d_w && g.ev) {       // an all-gather may still be reading this shard buffer
            HIPCHK(h, hipStreamWaitEvent(st, g.ev, 0));
            g.ptr = nullptr;
        }
    GevdParams p = base_params(h);
    if (ln == &h->lane[1] && h->d_Lspill_lane1) p.Lspill = h->d_Lspill_lane1;
    p.w = d_w;
    p.lam = d_lam;
    p.status = d_status;
    std::string why;
    hipError_t e;
    // order 64 takes the fused order-64 kernel in either arithmetic (kernels_gevd64.hip) unless that kernel is switched off
    if (split_f32_update(c)) {
        // large orders in f32: correlation on the f32 matrix cores into a scratch R (HBM round trip of
        // 2 L^2 c64 per bin, small against the eigen-iteration), then the LDS-resident GEVD from explicit R.  The scratch was
        // sized by apv_create (ensure_rscratch): nothing is allocated here.
        const size_t K = c.n_bins, L = c.n_srcs;
        if (!h->d_Rscratch || (2 * K * L * L + K * L) * 8 > h->rscratch_bytes) return fail(h, APV_ERR_STATE, "scratch R missing");
        float2* RB = (float2*)h->d_Rscratch;
        float2* RD = RB + K * L * L;
        float2* rr = RD + K * L * L;
        e = apv_launch_corr(APV_F32, c.n_bins, c.n_mics, c.n_srcs, (const float2*)d_XB, (const float2*)d_XD,
                            (const float2*)d_d, RB, RD, rr, st);
        if (e != hipSuccess) return hipfail(h, e, "corr launch");
        p.RB = RB;
        p.RD = RD;
        p.r = rr;
        e = apv_launch_gevd(p, APV_F32, false, st, &why);
    } else {
        p.XB = (const float2*)d_XB;
        p.XD = (const float2*)d_XD;
        p.d = (const float2*)d_d;
        e = apv_launch_gevd(p, h->cfg.compute_dtype, true, st, &why);
    }
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? APV_ERR_ARG : APV_ERR_HIP,
                                     why.empty() ? hipGetErrorString(e) : why);
    if (ln) {
        std::swap(ln->ev, ln->ev_prev);              // the latest becomes the one before; its handle is recorded anew
        ln->used_prev = ln->used;
        HIPCHK(h, hipEventRecord(ln->ev, st));
        ln->used = true;
    }
    return APV_OK;
}

int apv_set_update_streams(apv_handle* h, int32_t n) {
    if (!h) return APV_ERR_ARG;
    if (n != 1 && n != 2) return fail(h, APV_ERR_ARG, "update streams: 1 or 2");
    HIPCHK(h, hipSetDevice(h->device));
    // drain first: whatever is in flight was ordered under the old scheme
    if (int rc = lanes_sync(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n == 2 && !h->lane[0].s) {
        // Lane 0 IS the handle's control stream, lane 1 a stream of its own: the runtime deals a process's streams over four hardware
        // queues and every stream fewer is one fewer to share them with (with the gather's stream the sharded path then has three
        // streams and the lanes work beside the collective; with a control stream of its own they did not: DESIGN.md 4.9).  Waiting for
        // its own events costs the control stream nothing; a join (copies, timers) holds back lane 0's NEXT launch until lane 1's
        // latest has ended -- once per copy, not per launch.  APV_LANE0_CONTROL=0: a control stream apart from both lanes (A/B switch).
        static const bool lane0_ctrl = getenv("APV_LANE0_CONTROL") == nullptr || atoi(getenv("APV_LANE0_CONTROL")) != 0;
        for (auto& ln : h->lane) {
            if (lane0_ctrl && &ln == &h->lane[0]) ln.s = h->stream;
            else
            HIPCHK(h, hipStreamCreateWithFlags(&ln.s, hipStreamNonBlocking));
            HIPCHK(h, hipEventCreateWithFlags(&ln.ev, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&ln.ev_prev, hipEventDisableTiming));
        }
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    }
    if (n == 2 && h->lspill_bytes > 0 && !h->d_Lspill_lane1) HIPCHK(h, hipMalloc(&h->d_Lspill_lane1, h->lspill_bytes));
    for (auto& ln : h->lane) ln.used = ln.used_prev = ln.need_fork = false;
    h->ctrl_dirty = false;
    h->lane_next = 0;
    h->n_lanes = n;
    return APV_OK;
}

static int scan_status(apv_handle* h, const int32_t* st, int K) {
    int rc = APV_OK;
    for (int k = 0; k < K; ++k) {
        if (st[k] == 1) {
            char buf[96];
            std::snprintf(buf, sizeof(buf), "Matrix is not positive definite (bin %d)", k);
            return fail(h, APV_ERR_NOT_PD, buf);
        }
        if (st[k] == 2 && rc == APV_OK) {
            char buf[96];
            std::snprintf(buf, sizeof(buf), "eigen-iteration did not converge (bin %d)", k);
            h->err = buf;
            rc = APV_ERR_NO_CONVERGE;
        }
    }
    return rc;
}

int apv_update(apv_handle* h, const float* h_XB, const float* h_XD, const float* h_d, void* h_w, void* h_lam,
               int32_t* h_status) {
    if (!h) return APV_ERR_ARG;
    if (h->cfg.n_bins == 0) return APV_OK;
    if (!h_XB || !h_XD || !h_d || !h_w) return fail(h, APV_ERR_ARG, "null host pointer");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ensure_staging(h);
    if (rc != APV_OK) return rc;
    const apv_config& c = h->cfg;
    const size_t K = c.n_bins, M = c.n_mics, L = c.n_srcs;
    if ((rc = lanes_join(h, true)) != APV_OK) return rc;          // the staging buffers may still be read by a launch in flight
    HIPCHK(h, hipMemcpyAsync(h->d_XB, h_XB, K * M * L * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_XD, h_XD, K * M * L * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_d, h_d, K * M * 8, hipMemcpyHostToDevice, h->stream));
    rc = apv_update_dev(h, h->d_XB, h->d_XD, h->d_d, h->d_w, h->d_lam, h->d_status);
    if (rc != APV_OK) return rc;
    if ((rc = lanes_join(h, true)) != APV_OK) return rc;
    HIPCHK(h, hipMemcpyAsync(h_w, h->d_w, K * c.n_ranks * L * wsize(h), hipMemcpyDeviceToHost, h->stream));
    if (h_lam) HIPCHK(h, hipMemcpyAsync(h_lam, h->d_lam, K * L * lsize(h), hipMemcpyDeviceToHost, h->stream));
    std::string keep;
    int32_t* st = h_status;
    std::unique_ptr<int32_t[]> tmp;
    if (!st) {
        tmp.reset(new int32_t[K ? K : 1]);
        st = tmp.get();
    }
    HIPCHK(h, hipMemcpyAsync(st, h->d_status, K * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return scan_status(h, st, (int)K);
}

int apv_corr_dev(apv_handle* h, const void* d_XB, const void* d_XD, const void* d_d, void* d_RB, void* d_RD,
                 void* d_r) {
    if (!h || !d_XB || !d_XD || !d_d || !d_RB || !d_RD || !d_r) return fail(h, APV_ERR_ARG, "null device pointer");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;
    const apv_config& c = h->cfg;
    hipError_t e = apv_launch_corr(c.compute_dtype, c.n_bins, c.n_mics, c.n_srcs, (const float2*)d_XB,
                                   (const float2*)d_XD, (const float2*)d_d, d_RB, d_RD, d_r, h->stream);
    if (e != hipSuccess) return hipfail(h, e, "corr launch");
    return APV_OK;
}

int apv_to_bf16_dev(apv_handle* h, size_t count, const void* d_c64, void* d_bf16) {
    if (!h || !d_c64 || !d_bf16) return fail(h, APV_ERR_ARG, "null device pointer");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;
    hipError_t e = apv_launch_to_bf16(count, (const float2*)d_c64, (uint32_t*)d_bf16, h->stream);
    if (e != hipSuccess) return hipfail(h, e, "to_bf16 launch");
    return APV_OK;
}

int apv_corr_bf16_dev(apv_handle* h, const void* d_XB, const void* d_XD, const void* d_d, void* d_RB, void* d_RD,
                      void* d_r) {
    if (!h || !d_XB || !d_XD || !d_d || !d_RB || !d_RD || !d_r) return fail(h, APV_ERR_ARG, "null device pointer");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;
    const apv_config& c = h->cfg;
    if ((c.n_srcs != 32 && c.n_srcs != 64) || (c.n_mics % 16) != 0)
        return fail(h, APV_ERR_ARG, "bf16 correlation: n_srcs must be 32 or 64 and n_mics a multiple of 16");
    hipError_t e = apv_launch_corr_bf16(c.n_bins, c.n_mics, c.n_srcs, (const uint32_t*)d_XB, (const uint32_t*)d_XD,
                                        (const uint32_t*)d_d, (float2*)d_RB, (float2*)d_RD, (float2*)d_r, h->stream);
    if (e != hipSuccess) return hipfail(h, e, "corr_bf16 launch");
    return APV_OK;
}

int apv_gevd_vast_dev(apv_handle* h, const void* d_RB, const void* d_RD, const void* d_r, void* d_w, void* d_lam,
                      int32_t* d_status) {
    if (!h || !d_RB || !d_RD || !d_r || !d_w) return fail(h, APV_ERR_ARG, "null device pointer");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;
    GevdParams p = base_params(h);
    p.RB = d_RB;
    p.RD = d_RD;
    p.r = d_r;
    p.w = d_w;
    p.lam = d_lam;
    p.status = d_status;
    std::string why;
    hipError_t e = apv_launch_gevd(p, h->cfg.compute_dtype, false, h->stream, &why);
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? APV_ERR_ARG : APV_ERR_HIP,
                                     why.empty() ? hipGetErrorString(e) : why);
    return APV_OK;
}

int apv_jdiag_batched(apv_handle* h, int32_t n, int32_t batch, const double* h_A, const double* h_B, double* h_U,
                      double* h_lam, int32_t* h_status) {
    if (!h || !h_A || !h_B || !h_U || !h_lam) return fail(h, APV_ERR_ARG, "null host pointer");
    if (n < 1 || n > APV_MAX_N || batch < 0) return fail(h, APV_ERR_ARG, "jdiag order n must be in 1..64");
    if (batch == 0) return APV_OK;
    HIPCHK(h, hipSetDevice(h->device));
    // always runs in f64 / c128, whatever the handle's streaming dtype
    const size_t mat = (size_t)batch * n * n * 16;
    struct Tmp {
        void* p[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Tmp() { for (void* q : p) if (q) (void)hipFree(q); }
    } t;                                                    // freed on every way out
    void*& dA = t.p[0]; void*& dB = t.p[1]; void*& dU = t.p[2]; void*& dw = t.p[3]; void*& dl = t.p[4];
    void*& dsv = t.p[5]; void*& spill = t.p[6];
    HIPCHK(h, hipMalloc(&dA, mat));
    HIPCHK(h, hipMalloc(&dB, mat));
    HIPCHK(h, hipMalloc(&dU, mat));
    HIPCHK(h, hipMalloc(&dw, (size_t)batch * n * 16));
    HIPCHK(h, hipMalloc(&dl, (size_t)batch * n * 8));
    HIPCHK(h, hipMalloc(&dsv, (size_t)batch * sizeof(int32_t)));
    int32_t* ds = (int32_t*)dsv;
    const size_t sb = apv_gevd_spill_bytes(n, batch, APV_F64, h->cfg.reg_mode, 0.0, 1e-17, 1);   // as launched below: f64, tol 1e-17
    if (sb) HIPCHK(h, hipMalloc(&spill, sb));
    HIPCHK(h, hipMemcpyAsync(dA, h_A, mat, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dB, h_B, mat, hipMemcpyHostToDevice, h->stream));
    GevdParams p = base_params(h);
    p.n = n;
    p.K = batch;
    p.nV = 1;
    p.ranks[0] = 1;
    p.reg_bright = 0.0;
    p.sweep_tol2 = 1e-17;          // drop-in for jdiag: iterate to working precision
    p.out_c128 = 1;
    p.RB = dA;
    p.RD = dB;
    p.r = nullptr;
    p.w = dw;
    p.lam = dl;
    p.status = ds;
    p.U = dU;
    p.Lspill = spill;
    std::string why;
    hipError_t e = apv_launch_gevd(p, APV_F64, false, h->stream, &why);
    int rc = APV_OK;
    if (e != hipSuccess) rc = fail(h, APV_ERR_HIP, why.empty() ? hipGetErrorString(e) : why);
    std::unique_ptr<int32_t[]> tmp;
    int32_t* st = h_status;
    if (!st) {
        tmp.reset(new int32_t[batch]);
        st = tmp.get();
    }
    if (rc == APV_OK) {
        (void)hipMemcpyAsync(h_U, dU, mat, hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(h_lam, dl, (size_t)batch * n * 8, hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(st, ds, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        hipError_t se = hipStreamSynchronize(h->stream);
        if (se != hipSuccess) rc = hipfail(h, se, "jdiag sync");
    }
    if (rc != APV_OK) return rc;
    return scan_status(h, st, batch);
}

int apv_jdiag_large(apv_handle* h, int32_t n, int32_t batch, const double* h_A, const double* h_B, double* h_U,
                    double* h_lam, int32_t* h_status) {
    if (!h || !h_A || !h_B || !h_U || !h_lam) return fail(h, APV_ERR_ARG, "null host pointer");
    if (n < 1 || n > 2048 || batch < 0) return fail(h, APV_ERR_ARG, "apv_jdiag_large: n must be in 1..2048");
    if (batch == 0) return APV_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t mat = (size_t)batch * n * n * sizeof(double);
    struct Tmp {
        double* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Tmp() { for (double* q : p) if (q) (void)hipFree(q); }
    } t;                                                    // freed on every way out
    double*& dA = t.p[0]; double*& dB = t.p[1]; double*& dU = t.p[2]; double*& dl = t.p[3]; double*& dn = t.p[4];
    HIPCHK(h, hipMalloc((void**)&dA, mat));
    HIPCHK(h, hipMalloc((void**)&dB, mat));
    HIPCHK(h, hipMalloc((void**)&dU, mat));
    HIPCHK(h, hipMalloc((void**)&dl, (size_t)batch * n * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(dA, h_A, mat, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dB, h_B, mat, hipMemcpyHostToDevice, h->stream));
    std::unique_ptr<int32_t[]> tmp;
    int32_t* st = h_status;
    if (!st) {
        tmp.reset(new int32_t[batch]);
        st = tmp.get();
    }
    if (h->cfg.reg_mode == APV_REG_REL) {
        // B + reg_dark ||B||_2 I (apvast.py:26-27): the spectral norms first, four matrices per launch
        HIPCHK(h, hipMalloc((void**)&dn, (size_t)batch * sizeof(double)));
        for (int z0 = 0; z0 < batch; z0 += 4) {
            const double* mats[4];
            const int cnt = batch - z0 < 4 ? batch - z0 : 4;
            for (int q = 0; q < cnt; ++q) mats[q] = dB + (size_t)(z0 + q) * n * n;
            HIPCHK(h, apv_launch_norm2(n, cnt, mats, dn + z0, h->stream));
        }
    }
    int rc = apv_gevd_large(h, n, batch, dA, dB, h->cfg.reg_dark, dn, dU, dl, nullptr, 0.0, 0, nullptr, nullptr, st);
    if (rc == APV_OK || rc == APV_ERR_NOT_PD) {
        (void)hipMemcpyAsync(h_U, dU, mat, hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(h_lam, dl, (size_t)batch * n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
        (void)hipStreamSynchronize(h->stream);
    }
    if (rc != APV_OK) return rc;
    for (int z = 0; z < batch; ++z)
        if (st[z] == 2) return fail(h, APV_ERR_NO_CONVERGE, "eigen-iteration did not converge");
    return APV_OK;
}

// gathers the leading `rank` columns of U [batch][n][n] into [batch][n][rank]
__global__ void __launch_bounds__(256) lead_cols_kernel(int n, int rank, const double* __restrict__ U, double* __restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per = (size_t)n * rank;
    if (idx >= per) return;
    const int z = blockIdx.y;
    const int i = (int)(idx / rank), j = (int)(idx % rank);
    out[z * per + idx] = U[(size_t)z * n * n + (size_t)i * n + j];
}

int apv_jdiag_leading(apv_handle* h, int32_t n, int32_t batch, int32_t rank, const double* h_A, const double* h_B, double* h_U,
                      double* h_lam, int32_t* h_info) {
    if (!h || !h_A || !h_B || !h_U || !h_lam) return fail(h, APV_ERR_ARG, "null host pointer");
    if (n < 1 || n > 2048 || batch < 0) return fail(h, APV_ERR_ARG, "apv_jdiag_leading: n must be in 1..2048");
    if (rank < 1 || rank > n) return fail(h, APV_ERR_ARG, "apv_jdiag_leading: rank must be in 1..n");
    if (batch == 0) return APV_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t mat = (size_t)batch * n * n * sizeof(double);
    struct Tmp {
        double* p[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Tmp() { for (double* q : p) if (q) (void)hipFree(q); }
    } t;
    double*& dA = t.p[0]; double*& dB = t.p[1]; double*& dU = t.p[2]; double*& dl = t.p[3]; double*& dn = t.p[4]; double*& dV = t.p[5];
    HIPCHK(h, hipMalloc((void**)&dA, mat));
    HIPCHK(h, hipMalloc((void**)&dB, mat));
    HIPCHK(h, hipMalloc((void**)&dU, mat));
    HIPCHK(h, hipMalloc((void**)&dl, (size_t)batch * n * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&dV, (size_t)batch * n * rank * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(dA, h_A, mat, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(dB, h_B, mat, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.reg_mode == APV_REG_REL) {
        HIPCHK(h, hipMalloc((void**)&dn, (size_t)batch * sizeof(double)));
        for (int z0 = 0; z0 < batch; z0 += 4) {
            const double* mats[4];
            const int cnt = batch - z0 < 4 ? batch - z0 : 4;
            for (int q = 0; q < cnt; ++q) mats[q] = dB + (size_t)(z0 + q) * n * n;
            HIPCHK(h, apv_launch_norm2(n, cnt, mats, dn + z0, h->stream));
        }
    }
    std::vector<int32_t> st(batch, 0);
    h->gl_lead_rank = rank;
    int rc = apv_gevd_large(h, n, batch, dA, dB, h->cfg.reg_dark, dn, dU, dl, nullptr, 0.0, 0, nullptr, nullptr, st.data());
    const int lead_done = h->gl_lead_done;
    if (rc != APV_OK) return rc;
    const size_t per = (size_t)n * rank;
    hipLaunchKernelGGL(lead_cols_kernel, dim3((unsigned)((per + 255) / 256), batch), dim3(256), 0, h->stream, n, rank, dU, dV);
    HIPCHK(h, hipMemcpyAsync(h_U, dV, (size_t)batch * per * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    for (int z = 0; z < batch; ++z)
        HIPCHK(h, hipMemcpyAsync(h_lam + (size_t)z * rank, dl + (size_t)z * n, sizeof(double) * rank, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int z = 0; z < batch; ++z) {
        if (h_info) h_info[z] = lead_done ? 0 : 1;
        if (st[z] == 2) return fail(h, APV_ERR_NO_CONVERGE, "eigen-iteration did not converge");
    }
    return APV_OK;
}

int apv_stft_analysis_dev(apv_handle* h, int32_t n_ch, const float* d_x, void* d_spec) {
    if (!h || !d_x || !d_spec) return fail(h, APV_ERR_ARG, "null device pointer");
    if (h->cfg.block_size <= 0) return fail(h, APV_ERR_ARG, "handle was created without an STFT geometry");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;
    std::string why;
    hipError_t e = apv_launch_stft_analysis(h->cfg.block_size, n_ch, d_x, (float2*)d_spec, h->stream, &why);
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? APV_ERR_ARG : APV_ERR_HIP,
                                     why.empty() ? hipGetErrorString(e) : why);
    return APV_OK;
}

int apv_istft_ola_dev(apv_handle* h, int32_t n_ch, const void* d_spec, float* d_overlap, float* d_out) {
    if (!h || !d_spec || !d_overlap) return fail(h, APV_ERR_ARG, "null device pointer");
    if (h->cfg.block_size <= 0) return fail(h, APV_ERR_ARG, "handle was created without an STFT geometry");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_join(h, true)) return rc;
    std::string why;
    hipError_t e = apv_launch_istft_ola(h->cfg.block_size, h->cfg.hop_size, n_ch, (const float2*)d_spec, d_overlap,
                                        d_out, h->stream, &why);
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? APV_ERR_ARG : APV_ERR_HIP,
                                     why.empty() ? hipGetErrorString(e) : why);
    return APV_OK;
}

int apv_comm_unique_id(char id_out[128]) {
    if (!id_out) return APV_ERR_ARG;
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (ncclGetUniqueId(&id) != ncclSuccess) return APV_ERR_RCCL;
    std::memcpy(id_out, &id, 128);
    return APV_OK;
}

int apv_comm_init(apv_handle* h, const char id[128], int32_t rank, int32_t world) {
    if (!h || !id || world < 1 || rank < 0 || rank >= world) return fail(h, APV_ERR_ARG, "bad communicator arguments");
    HIPCHK(h, hipSetDevice(h->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, 128);
    ncclComm_t comm;
    ncclResult_t r = ncclCommInitRank(&comm, world, uid, rank);
    if (r != ncclSuccess) return fail(h, APV_ERR_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    // the communicator must be the one that was asked for: a rank that joined another job's id, or a world of another size,
    // would gather somebody else's shards without any call failing
    int count = -1, urank = -1;
    if (ncclCommCount(comm, &count) != ncclSuccess || ncclCommUserRank(comm, &urank) != ncclSuccess || count != world || urank != rank) {
        (void)ncclCommDestroy(comm);
        return fail(h, APV_ERR_RCCL, "communicator reports " + std::to_string(count) + " ranks / rank " + std::to_string(urank) +
                                         ", expected " + std::to_string(world) + " / " + std::to_string(rank));
    }
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_world = world;
    if (!h->comm_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
        for (auto& g : h->gather_done) HIPCHK(h, hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
        HIPCHK(h, hipEventCreate(&h->ev_ag0));
        HIPCHK(h, hipEventCreate(&h->ev_ag1));
        HIPCHK(h, hipMalloc((void**)&h->d_bar, sizeof(int32_t)));
        HIPCHK(h, hipMemset(h->d_bar, 0, sizeof(int32_t)));
    }
    return APV_OK;
}

int apv_allgather_filters_dev(apv_handle* h, const void* d_w_shard, void* d_w_all) {
    if (!h || !d_w_shard || !d_w_all) return fail(h, APV_ERR_ARG, "null device pointer");
    if (!h->comm) return fail(h, APV_ERR_RCCL, "communicator not initialised (apv_comm_init)");
    const apv_config& c = h->cfg;
    const size_t bytes = (size_t)c.n_bins * c.n_ranks * c.n_srcs * wsize(h);
    HIPCHK(h, hipSetDevice(h->device));
    // the gather starts once the kernels already queued on the compute stream have written the shard, and runs
    // on its own stream: the next block's update (into another shard buffer) overlaps it
    if (lanes_on(h)) {
        // the shard was written on a lane: the gather waits for the latest launch of BOTH lanes (whichever wrote it, now or
        // earlier; the other one's finished a step ago in the usual order of calls) -- on its own stream, so that no stream a
        // later launch could queue up behind carries the wait -- and for the control stream as below
        for (auto& ln : h->lane)
            if (ln.used) HIPCHK(h, hipStreamWaitEvent(h->comm_stream, ln.ev, 0));
    }
    HIPCHK(h, hipEventRecord(h->ev_ready, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_ready, 0));
    HIPCHK(h, hipEventRecord(h->ev_ag0, h->comm_stream));
    ncclResult_t r = ncclAllGather(d_w_shard, d_w_all, bytes, ncclChar, (ncclComm_t)h->comm, h->comm_stream);
    if (r != ncclSuccess) return fail(h, APV_ERR_RCCL, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    HIPCHK(h, hipEventRecord(h->ev_ag1, h->comm_stream));
    h->ag_bytes = bytes;
    // remember "this shard buffer is being read until <event>": the slot already tracking this buffer, else a free one,
    // else the oldest -- whose gather the compute stream then waits for here, so no later update can overtake it
    int slot = -1;
    for (int i = 0; i < 4 && slot < 0; ++i)
        if (h->gather_done[i].ptr == d_w_shard) slot = i;
    for (int i = 0; i < 4 && slot < 0; ++i)
        if (h->gather_done[i].ptr == nullptr) slot = i;
    if (slot < 0) {
        slot = h->gather_next;
        h->gather_next = (h->gather_next + 1) & 3;
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->gather_done[slot].ev, 0));
        if (lanes_on(h))
            for (auto& ln : h->lane) HIPCHK(h, hipStreamWaitEvent(ln.s, h->gather_done[slot].ev, 0));
    }
    h->gather_done[slot].ptr = d_w_shard;
    HIPCHK(h, hipEventRecord(h->gather_done[slot].ev, h->comm_stream));
    return APV_OK;
}

int apv_host_alloc(void** p, size_t bytes) {
    if (!p) return APV_ERR_ARG;
    *p = nullptr;
    if (hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        return APV_ERR_HIP;
    }
    apv_host_blocks_note(*p, bytes ? bytes : 1, true);
    return APV_OK;
}

int apv_host_free(void* p) {
    if (!p) return APV_OK;
    apv_host_blocks_note(p, 0, false);
    return hipHostFree(p) == hipSuccess ? APV_OK : APV_ERR_HIP;
}

int apv_comm_count(apv_handle* h, int32_t* n_ranks, int32_t* user_rank) {
    if (!h || !n_ranks) return fail(h, APV_ERR_ARG, "null argument");
    if (!h->comm) return fail(h, APV_ERR_RCCL, "communicator not initialised (apv_comm_init)");
    int count = 0, urank = 0;
    ncclResult_t r = ncclCommCount((ncclComm_t)h->comm, &count);
    if (r == ncclSuccess) r = ncclCommUserRank((ncclComm_t)h->comm, &urank);
    if (r != ncclSuccess) return fail(h, APV_ERR_RCCL, std::string("ncclCommCount: ") + ncclGetErrorString(r));
    *n_ranks = count;
    if (user_rank) *user_rank = urank;
    return APV_OK;
}

int apv_comm_last_gather(apv_handle* h, float* elapsed_ms, size_t* bytes_per_rank) {
    if (!h || !elapsed_ms) return fail(h, APV_ERR_ARG, "null argument");
    if (!h->comm || h->ag_bytes == 0) return fail(h, APV_ERR_RCCL, "no all-gather has run on this handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(h->ev_ag1));
    HIPCHK(h, hipEventElapsedTime(elapsed_ms, h->ev_ag0, h->ev_ag1));
    if (bytes_per_rank) *bytes_per_rank = h->ag_bytes;
    return APV_OK;
}

int apv_comm_barrier(apv_handle* h) {
    if (!h) return APV_ERR_ARG;
    if (!h->comm) return fail(h, APV_ERR_RCCL, "communicator not initialised (apv_comm_init)");
    HIPCHK(h, hipSetDevice(h->device));
    if (int rc = lanes_sync(h)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    ncclResult_t r = ncclAllReduce(h->d_bar, h->d_bar, 1, ncclInt32, ncclSum, (ncclComm_t)h->comm, h->comm_stream);
    if (r != ncclSuccess) return fail(h, APV_ERR_RCCL, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    HIPCHK(h, hipStreamSynchronize(h->comm_stream));
    return APV_OK;
}

int apv_debug_set_stamps(apv_handle* h, void* d_stamps) {
    if (!h) return APV_ERR_ARG;
    h->d_stamps = static_cast<unsigned long long*>(d_stamps);
    return APV_OK;
}

int apv_device_sync(apv_handle* h) {
    if (!h) return APV_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    return APV_OK;
}

int apv_device_info(apv_handle* h, char name_out[128], int32_t* n_cus, int32_t* clock_mhz) {
    if (!h || !name_out) return APV_ERR_ARG;
    hipDeviceProp_t prop;
    HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
    std::snprintf(name_out, 128, "%s (%s)", prop.name, prop.gcnArchName);
    if (n_cus) *n_cus = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    return APV_OK;
}

}  // extern "C"
