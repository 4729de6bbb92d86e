// Shared host/device infrastructure of libmxf_gp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <stdlib.h>
#include "../../include/mxf_gp.h"

// Tuning knobs.  The shipped library is built WITHOUT -DMXF_PROBES: every knob is the compile-time default (the measured best) and the
// library reads no MXF_* environment variable except MXF_RCCL_LIB (a path).  The probe build (make probe -> libmxf_gp_probe.so, selected
// with MXF_GP_LIB by tests/probes/*) reads them from the environment once per process, for A/B measurements.
#ifdef MXF_PROBES
#define MXF_KNOB(name, dflt) (getenv(name) ? atoll(getenv(name)) : (long long)(dflt))
#define MXF_KNOB_SET(name) (getenv(name) != nullptr)
#else
#define MXF_KNOB(name, dflt) ((long long)(dflt))
#define MXF_KNOB_SET(name) (false)
#endif

// Probe build only (MXF_SVGP_STAGES=1): device-time stamps of the SVGP training call's stages, printed (stderr) at the end of the call after a
// device synchronise -- the call's own critical path without a profiler in the way (the profiler's ~30 us per launch makes a 4-sample step
// host-bound and its timeline misleading).  tests/probes/svgp_stages.py
#ifdef MXF_PROBES
struct mxf_stage_log {
    static const int N = 48;
    hipEvent_t ev[N]; const char* name[N]; int n = 0; bool made = false;
    void mark(const char* nm, hipStream_t s) {
        if (!made) { for (int i = 0; i < N; ++i) (void)hipEventCreate(&ev[i]); made = true; }
        if (n < N) { name[n] = nm; (void)hipEventRecord(ev[n], s); ++n; }
    }
    void dump() {
        (void)hipDeviceSynchronize();
        for (int i = 1; i < n; ++i) { float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[0], ev[i]); fprintf(stderr, "  stage %-28s %8.3f ms\n", name[i], ms); }
        fprintf(stderr, "  --\n");
        n = 0;
    }
};
#define MXF_STAGE(h, nm, s) do { static const bool on_ = MXF_KNOB("MXF_SVGP_STAGES", 0) != 0; if (on_) (h)->stages.mark(nm, s); } while (0)
#define MXF_STAGE_DUMP(h) do { static const bool on_ = MXF_KNOB("MXF_SVGP_STAGES", 0) != 0; if (on_) (h)->stages.dump(); } while (0)
#else
#define MXF_STAGE(h, nm, s) do { } while (0)
#define MXF_STAGE_DUMP(h) do { } while (0)
#endif

// In-step durations of the SVGP training call's bulk kernels (mxf_svgp_timing): HIP events around them on the stream each runs on -- what
// bench.py's roofline entries for the in-step passes divide by.  Off by default (two event records per kernel when on).
constexpr int MXF_NT = 8;
struct mxf_timing {
    bool on = false, made = false;
    hipEvent_t ev[2 * MXF_NT];
    bool used[MXF_NT] = {false, false, false, false, false, false, false, false};
    bool init() {
        if (made) return true;
        for (int i = 0; i < 2 * MXF_NT; ++i) if (hipEventCreate(&ev[i]) != hipSuccess) return false;
        made = true;
        return true;
    }
};
enum { MXF_T_PLANES_A = 0, MXF_T_PSI2 = 1, MXF_T_PLANES_B = 2, MXF_T_TGEMM = 3, MXF_T_BWD = 4, MXF_T_CHAIN = 5, MXF_T_VGEMM = 6, MXF_T_CALL = 7 };
#define MXF_T0(h, i, s) do { if ((h)->tm.on) { (void)hipEventRecord((h)->tm.ev[2 * (i)], s); (h)->tm.used[i] = true; } } while (0)
#define MXF_T1(h, i, s) do { if ((h)->tm.on) (void)hipEventRecord((h)->tm.ev[2 * (i) + 1], s); } while (0)

struct mxf_ctx {
    mxf_timing tm;
#ifdef MXF_PROBES
    mxf_stage_log stages;
#endif
    int device = 0;
    std::string err;
    void* ws = nullptr;     // scratch, grown on demand (hipMalloc; never inside graph capture)
    size_t ws_bytes = 0;
    hipStream_t side = nullptr;   // internal side streams: independent chains of the SVGP step run concurrently
    hipStream_t side2 = nullptr;
    hipStream_t potrf_aux = nullptr;                        // look-ahead stream of the blocked Cholesky (chol.hip)
    bool potrf_aux_ready = false;                           // set only when EVERY auxiliary stream and event below exists
    hipEvent_t ev_pa = nullptr, ev_pb = nullptr, ev_ph = nullptr;
    hipStream_t potrf_inv = nullptr;                        // r05: the inverse of the factor, row block by row block NEXT TO the factorisation (chol.hip)
    hipEvent_t ev_pi = nullptr, ev_pj = nullptr;
    hipStream_t potrf_rows = nullptr;                       // r06: the rows FAR below an outer panel are solved here, next to the next panel's chain
    hipEvent_t ev_pc = nullptr, ev_rb = nullptr;
    hipStream_t potrf_acc = nullptr;                        // r06: kcoef L^-T L^-1 accumulated row block by row block of the eager inverse
    hipEvent_t ev_pq = nullptr, ev_pz = nullptr;
    hipStream_t potrf_chain = nullptr;                      // r06 (MXF_POTRF_CUMASK): the factorisation's own chain stream when the bulk streams are CU-masked
    hipEvent_t ev_pk = nullptr;
    bool potrf_chain_always = false;
    bool potrf_masked = false;                              // potrf_aux / potrf_inv were created with a CU mask (blocking streams: see mxf_potrf_aux_init)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_aux = nullptr, ev_aux2 = nullptr, ev_su = nullptr;
    hipEvent_t ev_k1 = nullptr, ev_k2 = nullptr, ev_k3 = nullptr;   // r06: the SVGP call's condition norms and value scalars leave the caller's stream (composite.hip)
    hipEvent_t ev_tg = nullptr;       // the T product has been enqueued / finished (whitened few-sample form: Phi runs behind it)
    void* gram_ws = nullptr;   // pre-scaled coordinates of mxf_gram (separate: composites hold `ws` while calling mxf_gram)
    size_t gram_ws_bytes = 0;
    int64_t ws_generation = 0; // bumped whenever `ws` / `gram_ws` is freed and re-allocated: device pointers baked into a captured hipGraph are stale after that
    double* cond_dev = nullptr; // [ |Kuu + jitter I|_1, |(Kuu + jitter I)^-1|_1 ] of the last SVGP training call (mxf_svgp_last_cond)
    double* cond_host = nullptr; // pinned, device-visible host words, MXF_COND_SLOTS x [running MAX, last] of the condition numbers the training calls published into their slot (mxf_svgp_cond_nowait / mxf_svgp_cond_slot)
    int svgp_form = 0;         // float32 streaming form of the next SVGP training calls (mxf_svgp_configure): 0 explicit inverse, 1 whitened
    int cond_slot = 0;         // the slot the next SVGP training calls publish their condition number into
    void* bwd_acc = nullptr;   // scratch of the MFMA reverse pass (gram_bwd.hip): float64 row-side sums [M][16] + 16, scaled coordinates
    size_t bwd_acc_bytes = 0;
    void* comm = nullptr;      // RCCL communicator of mxf_comm_init (comm.hip); nullptr on single-GPU handles
    int comm_nranks = 0, comm_rank = -1;
    double* pinv = nullptr;    // ring of 16 x 16 diagonal-block inverses handed from the factoring to the solving workgroups of potrf_tiles_kernel
    size_t pinv_elems = 0, pinv_cursor = 0;
    int* flags = nullptr;      // zero-initialised arrival counters for in-kernel workgroup hand-offs (potrf panel); each use leaves 0 behind
    unsigned flag_cursor = 0;
    unsigned* gsync = nullptr; // zero-initialised rendezvous counters of the wide split GEMMs (pacing hints only; gemm_split.hip); each use leaves 0 behind
    unsigned gsync_cursor = 0;
};
constexpr int MXF_COND_SLOTS = 64;
// the condition words of the SVGP training call: device accumulators + the pinned host slots (allocated on first use)
static inline bool mxf_cond_init(mxf_ctx* h) {
    if (!h->cond_dev) {
        if (hipMalloc((void**)&h->cond_dev, 4 * sizeof(double)) != hipSuccess) { h->cond_dev = nullptr; return false; }
        if (hipMemset(h->cond_dev, 0, 4 * sizeof(double)) != hipSuccess) return false;
    }
    if (!h->cond_host) {
        if (hipHostMalloc((void**)&h->cond_host, 2 * MXF_COND_SLOTS * sizeof(double), hipHostMallocMapped) != hipSuccess) { h->cond_host = nullptr; return false; }
        for (int i = 0; i < 2 * MXF_COND_SLOTS; ++i) h->cond_host[i] = 0.0;
    }
    return true;
}
constexpr unsigned MXF_NGSYNC = 1u << 18;
// a fresh run of `count` zeroed rendezvous counters (rotating: a run is reused only after 2^18 / count later launches have been queued --
// by then the launch that used it has long left them at zero); nullptr = none available (the caller then launches without rendezvous)
static inline unsigned* mxf_gsync(mxf_ctx* h, unsigned count) {
    if (count == 0 || count > MXF_NGSYNC / 4) return nullptr;
    if (!h->gsync) {
        if (hipMalloc((void**)&h->gsync, MXF_NGSYNC * sizeof(unsigned)) != hipSuccess) { h->gsync = nullptr; return nullptr; }
        if (hipMemset(h->gsync, 0, MXF_NGSYNC * sizeof(unsigned)) != hipSuccess) return nullptr;
    }
    if (h->gsync_cursor + count > MXF_NGSYNC) h->gsync_cursor = 0;
    unsigned* p = h->gsync + h->gsync_cursor;
    h->gsync_cursor += count;
    return p;
}
constexpr unsigned MXF_NFLAGS = 1u << 18;
// a fresh run of `count` zeroed counters (rotating; a slot is reused only after 2^18 / count later launches have been queued)
static inline int* mxf_flags(mxf_ctx* h, unsigned count) {
    if (!h->flags) {
        if (hipMalloc((void**)&h->flags, MXF_NFLAGS * sizeof(int)) != hipSuccess) { h->flags = nullptr; return nullptr; }
        if (hipMemset(h->flags, 0, MXF_NFLAGS * sizeof(int)) != hipSuccess) return nullptr;
    }
    if (count > MXF_NFLAGS) return nullptr;
    if (h->flag_cursor + count > MXF_NFLAGS) h->flag_cursor = 0;
    int* p = h->flags + h->flag_cursor;
    h->flag_cursor += count;
    return p;
}

// ring allocator of the Cholesky tile kernel's inverse blocks: the ring holds at least two regions of the largest request, so a region is
// reused no earlier than the launch after next ON THE SAME STREAM -- by then its readers (the previous launch) have finished in stream order.
// (Two streams factoring through ONE handle would share the ring: the C ABI's rule is one handle per thread and calls not re-entrant.)
static inline double* mxf_potrf_inv(mxf_ctx* h, size_t elems) {
    if (elems * 2 > h->pinv_elems) {
        if (h->pinv) { (void)hipDeviceSynchronize(); (void)hipFree(h->pinv); h->pinv = nullptr; h->pinv_elems = 0; }
        size_t want = elems * 4 > ((size_t)4 << 20) ? elems * 4 : ((size_t)4 << 20);      // >= 32 MB
        if (hipMalloc((void**)&h->pinv, want * sizeof(double)) != hipSuccess) { h->pinv = nullptr; return nullptr; }
        h->pinv_elems = want; h->pinv_cursor = 0;
        ++h->ws_generation;
    }
    if (h->pinv_cursor + elems > h->pinv_elems) h->pinv_cursor = 0;
    double* p = h->pinv + h->pinv_cursor;
    h->pinv_cursor += elems;
    return p;
}

#define MXF_FAIL(h, code, ...)                                   \
    do {                                                         \
        char _b[512];                                            \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
        if (h) (h)->err = _b;                                    \
        return (code);                                           \
    } while (0)

#define MXF_HIP(h, call)                                                                   \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess)                                                              \
            MXF_FAIL(h, -100 - (int)_e, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
    } while (0)

#define MXF_LAUNCH_CHECK(h)                                                                \
    do {                                                                                   \
        hipError_t _e = hipGetLastError();                                                 \
        if (_e != hipSuccess)                                                              \
            MXF_FAIL(h, -100 - (int)_e, "%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
    } while (0)

// scratch allocator: returns a pointer valid until the next call that needs more
static inline void* mxf_ws(mxf_ctx* h, size_t bytes) {
    if (bytes <= h->ws_bytes) return h->ws;
    if (h->ws) {
        (void)hipDeviceSynchronize();
        (void)hipFree(h->ws);
        h->ws = nullptr;
        h->ws_bytes = 0;
    }
    ++h->ws_generation;
    size_t want = bytes + (bytes >> 2) + (1u << 20);
    if (hipMalloc(&h->ws, want) != hipSuccess) {
        h->ws = nullptr;
        return nullptr;
    }
    h->ws_bytes = want;
    return h->ws;
}

static inline bool mxf_potrf_aux_init(mxf_ctx* h) {
    if (h->potrf_aux_ready) return true;
    // all or nothing: a partial failure leaves NO auxiliary stream / event behind (the callers gate look-ahead on the return value)
    // MXF_POTRF_CUMASK = c > 0: the two streams that carry the bulk products next to the factorisation (trailing updates, eager inverse)
    // are created with a CU mask that leaves c CUs of every XCD to the caller's stream -- the latency chain's few workgroups (76 KB of LDS
    // each) cannot share a CU with a 133 KB product workgroup and otherwise queue behind whole product workgroups.  The excluded set
    // {32 x + (x + 8 j) % 32 : x < 8, j < c} holds c CUs per XCD whether mask bit i means (XCD i % 8, CU i / 8) or (XCD i / 32, CU i % 32).
    static const int cumask_env = (int)MXF_KNOB("MXF_POTRF_CUMASK", 0);
    auto make_stream = [&](hipStream_t* s_) -> bool {
        if (cumask_env > 0 && cumask_env < 32) {
            uint32_t mask[8];
            for (int x = 0; x < 8; ++x) {
                mask[x] = 0xffffffffu;
                for (int j = 0; j < cumask_env; ++j) mask[x] &= ~(1u << ((x + 8 * (j % 4) + (j / 4)) % 32));
            }
            if (hipExtStreamCreateWithCUMask(s_, 8, mask) == hipSuccess) { h->potrf_masked = true; return true; }
            (void)hipGetLastError();
        }
        return hipStreamCreateWithFlags(s_, hipStreamNonBlocking) == hipSuccess;
    };
    bool ok = make_stream(&h->potrf_aux);
    if (!ok) h->potrf_aux = nullptr;
    ok = ok && make_stream(&h->potrf_inv);
    if (!ok) h->potrf_inv = nullptr;
    ok = ok && hipStreamCreateWithFlags(&h->potrf_rows, hipStreamNonBlocking) == hipSuccess;
    if (!ok) h->potrf_rows = nullptr;
    // CU-masked streams are created WITHOUT hipStreamNonBlocking (the API has no flags): they synchronise with the legacy default stream,
    // which is PyTorch's default stream -- so in that mode the factorisation's chain runs on a non-blocking stream of its own, forked from
    // and joined to the caller's stream, and nothing is launched on the caller's stream in between
    // MXF_POTRF_CHAIN_PRIO=1 (probe): the chain on a most-urgent stream of its own even without the masks
    static const int chain_prio_env = (int)MXF_KNOB("MXF_POTRF_CHAIN_PRIO", 0);
    // (both experiment streams exist only when their probe knob asks for them: the product build never creates them)
    if (chain_prio_env) {
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
        ok = ok && hipStreamCreateWithPriority(&h->potrf_chain, hipStreamNonBlocking, hi) == hipSuccess;
        if (ok) h->potrf_chain_always = true;
    } else if (cumask_env > 0)
    ok = ok && hipStreamCreateWithFlags(&h->potrf_chain, hipStreamNonBlocking) == hipSuccess;
    if (!ok) h->potrf_chain = nullptr;
    static const int kacc_env_ = (int)MXF_KNOB("MXF_POTRF_KACC", 0);
    if (kacc_env_) ok = ok && make_stream(&h->potrf_acc);
    if (!ok) h->potrf_acc = nullptr;
    hipEvent_t* evs[] = {&h->ev_pa, &h->ev_pb, &h->ev_ph, &h->ev_pi, &h->ev_pj, &h->ev_pc, &h->ev_rb, &h->ev_pk, &h->ev_pq, &h->ev_pz};
    for (hipEvent_t* e : evs)
        if (ok && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { *e = nullptr; ok = false; }
    if (!ok) {
        for (hipEvent_t* e : evs) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
        if (h->potrf_acc) { (void)hipStreamDestroy(h->potrf_acc); h->potrf_acc = nullptr; }
        if (h->potrf_chain) { (void)hipStreamDestroy(h->potrf_chain); h->potrf_chain = nullptr; }
        if (h->potrf_rows) { (void)hipStreamDestroy(h->potrf_rows); h->potrf_rows = nullptr; }
        if (h->potrf_inv) { (void)hipStreamDestroy(h->potrf_inv); h->potrf_inv = nullptr; }
        if (h->potrf_aux) { (void)hipStreamDestroy(h->potrf_aux); h->potrf_aux = nullptr; }
        return false;
    }
    h->potrf_aux_ready = true;
    return true;
}

static inline bool mxf_side_init(mxf_ctx* h) {
    if (h->side) return true;
    // (a CU-masked side2 via hipExtStreamCreateWithCUMask was measured: 83 -> 114 ms per step; stream priorities -- bulk stream lowest,
    //  Su-chain stream highest -- were measured in r01 and again in r05 (probe knob MXF_STREAM_PRIO=1); plain streams kept)
    static const int prio_env = (int)MXF_KNOB("MXF_STREAM_PRIO", 0);
    int lo = 0, hi = 0;
    if (prio_env && hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }      // lo = least urgent (numerically largest)
    if (prio_env) {
        if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo) != hipSuccess) { h->side = nullptr; return false; }
        // (1: the Su chain's stream most urgent; 2: BOTH side streams least urgent -- the caller's stream carries the Kuu chain at normal priority)
        if (hipStreamCreateWithPriority(&h->side2, hipStreamNonBlocking, prio_env == 2 ? lo : hi) != hipSuccess) { h->side2 = nullptr; return false; }
    } else {
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) { h->side = nullptr; return false; }
    if (hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking) != hipSuccess) { h->side2 = nullptr; return false; }
    }
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_aux, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_aux2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_tg, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_su, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_k1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_k2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_k3, hipEventDisableTiming) != hipSuccess) return false;
    return true;
}

static inline void* mxf_gram_ws(mxf_ctx* h, size_t bytes) {
    if (bytes <= h->gram_ws_bytes) return h->gram_ws;
    if (h->gram_ws) {
        (void)hipDeviceSynchronize();
        (void)hipFree(h->gram_ws);
        h->gram_ws = nullptr;
        h->gram_ws_bytes = 0;
    }
    ++h->ws_generation;
    size_t want = bytes + (bytes >> 2) + (1u << 16);
    if (hipMalloc(&h->gram_ws, want) != hipSuccess) { h->gram_ws = nullptr; return nullptr; }
    h->gram_ws_bytes = want;
    return h->gram_ws;
}

static inline size_t mxf_esize(int dtype) { return dtype == MXF_F64 ? 8 : 4; }
static inline size_t mxf_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- device helpers -------------------------------------------------------------------------
template <typename T> struct Vec16;   // 16-byte vector of T
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef double f64x2_t __attribute__((ext_vector_type(2)));
template <> struct Vec16<float> { typedef f32x4_t type; static constexpr int n = 4; };
template <> struct Vec16<double> { typedef f64x2_t type; static constexpr int n = 2; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// wave64 sum with DPP (no LDS traffic); the total is valid in LANE 63 only.
// quad butterflies, row_half_mirror, row_mirror give every lane of a 16-lane row its row sum; row_bcast:15 / :31 fold rows.
#define MXF_DPP_F(v, ctrl, rmask) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ float wave_sum63(float v) {
    v += MXF_DPP_F(v, 0xB1, 0xf);    // quad_perm [1,0,3,2]
    v += MXF_DPP_F(v, 0x4E, 0xf);    // quad_perm [2,3,0,1]
    v += MXF_DPP_F(v, 0x141, 0xf);   // row_half_mirror
    v += MXF_DPP_F(v, 0x140, 0xf);   // row_mirror
    v += MXF_DPP_F(v, 0x142, 0xa);   // row_bcast:15 -> rows 1,3
    v += MXF_DPP_F(v, 0x143, 0xc);   // row_bcast:31 -> rows 2,3
    return v;
}
#define MXF_DPP_D(v, ctrl, rmask)                                                                                  \
    __builtin_bit_cast(double, (((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(                        \
                                     0, (int)(__builtin_bit_cast(unsigned long long, (v)) >> 32), (ctrl), (rmask), 0xf, false)) << 32) | \
                                (unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(                         \
                                    0, (int)(__builtin_bit_cast(unsigned long long, (v)) & 0xffffffffull), (ctrl), (rmask), 0xf, false))
__device__ __forceinline__ double wave_sum63(double v) {
    v += MXF_DPP_D(v, 0xB1, 0xf);
    v += MXF_DPP_D(v, 0x4E, 0xf);
    v += MXF_DPP_D(v, 0x141, 0xf);
    v += MXF_DPP_D(v, 0x140, 0xf);
    v += MXF_DPP_D(v, 0x142, 0xa);
    v += MXF_DPP_D(v, 0x143, 0xc);
    return v;
}

// ---- row-level reduce-scatter of N per-lane values (N = 2, 4, 8, 16) -------------------------------------------------------------
// Input: every lane holds a[0..N-1].  Output: lane l of each 16-lane row holds  sum over the row's 16 lanes of a[l & (N-1)]
// (all lanes valid; lanes with (l & 15) < N form one complete set per row).  A halving butterfly: at each level a lane keeps one
// half of its values and receives the partner's partial sums of that half, so the cost is (N-1) DPP adds + 2(N-1) selects
// instead of 4N DPP adds for N separate row reductions.  Partners (row_mirror, row_half_mirror, quad reverse, quad xor-1) always hold
// the same half, so every level is ONE symmetric DPP move.
template <int CTRL> __device__ __forceinline__ float mxf_dpp_mov(float v) { return MXF_DPP_F(v, CTRL, 0xf); }
template <int CTRL> __device__ __forceinline__ double mxf_dpp_mov(double v) { return MXF_DPP_D(v, CTRL, 0xf); }

template <typename T, int HALF, int CTRL>
__device__ __forceinline__ void mxf_rs_level(T* a, bool upper) {
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const T keep = upper ? a[i + HALF] : a[i];
        const T send = upper ? a[i] : a[i + HALF];
        a[i] = keep + mxf_dpp_mov<CTRL>(send);
    }
}
template <typename T, int N>
__device__ __forceinline__ T row_reduce_scatter(T* a /* N values, clobbered */, int lane) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16, "row_reduce_scatter: N must be a power of two <= 16");
    if (N >= 16) mxf_rs_level<T, 8, 0x140>(a, (lane & 8) != 0);                 // row_mirror        i <-> 15-i
    if (N >= 8) mxf_rs_level<T, (N >= 8 ? 4 : 1), 0x141>(a, (lane & 4) != 0);   // row_half_mirror   i <-> 7-i (within 8)
    if (N >= 4) mxf_rs_level<T, (N >= 4 ? 2 : 1), 0x1B>(a, (lane & 2) != 0);    // quad_perm [3,2,1,0]
    if (N >= 2) mxf_rs_level<T, 1, 0xB1>(a, (lane & 1) != 0);                   // quad_perm [1,0,3,2]
    T v = a[0];
    // lanes that hold the same index differ in the bits above log2(N): fold them with rotations (index-preserving)
    if (N <= 1) v += mxf_dpp_mov<0xB1>(v);
    if (N <= 2) v += mxf_dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]  (xor 2)
    if (N <= 4) v += mxf_dpp_mov<0x124>(v);     // row_ror:4
    if (N <= 8) v += mxf_dpp_mov<0x128>(v);     // row_ror:8
    return v;
}

// lane-wise sum over the wave's four 16-lane rows (every lane ends with x[l%16 of row0] + ... + x[l%16 of row3]): the gfx950
// v_permlane16_swap / v_permlane32_swap instructions.  Inline asm: the clang builtin (ROCm 7.2) folds its two results into one
// register and returns 2x (tests/probes/probe_permlane.hip).
__device__ __forceinline__ float wave_rows_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    a = a + b; b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in thread 0
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 16 */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    T r = 0;
    if (threadIdx.x < 64) {
        r = (threadIdx.x < nw) ? smem[threadIdx.x] : (T)0;
        r = wave_sum(r);
    }
    return r;
}

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// C-ABI entry points implemented across translation units share these internal (typed) launchers

// ---- float64 exponentials of the Gram kernels (arguments are never positive) ------------------------------------------------------
// The device library's exp / exp2 carry special-case handling the covariance functions never need; these are the bare forms:
// round-to-nearest split, degree-13 polynomial on |f| <= 1/2 (2^f) resp. |r| <= ln2/2 (e^r), v_ldexp_f64 (which also gives the gradual
// underflow).  Max error 1 ulp against libm over [0, 1100] (checked on the host with the same fma sequence).  RBF Gram float64 at
// N=65536: 7.43 -> 7.06 ms.  (Replacing v_rndne / v_cvt / v_ldexp by the 1.5 * 2^52 trick and an exponent-field add was slower: 7.73 ms.)
__device__ __forceinline__ double mxf_exp2_neg_f64(double x) {     // 2^(-x), x >= 0
    const double y = -x, n = __builtin_rint(y), f = y - n;
    double p = 1.369148885390412888e-12;
    p = fma(p, f, 2.567843599348820514e-11); p = fma(p, f, 4.445538271870811498e-10); p = fma(p, f, 7.054911620801123329e-09);
    p = fma(p, f, 1.017808600923969973e-07); p = fma(p, f, 1.321548679014430949e-06); p = fma(p, f, 1.525273380405984028e-05);
    p = fma(p, f, 0.0001540353039338160995); p = fma(p, f, 0.001333355814642844342); p = fma(p, f, 0.009618129107628477162);
    p = fma(p, f, 0.05550410866482157995); p = fma(p, f, 0.2402265069591007123); p = fma(p, f, 0.6931471805599453094);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)n);
}
__device__ __forceinline__ double mxf_exp_nonpos_f64(double x) {   // e^x, x <= 0
    const double xc = fmax(x, -800.0), n = __builtin_rint(xc * 1.4426950408889634074);
    double r = fma(-n, 0.693147180369123816490, xc);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0); p = fma(p, r, 1.0 / 39916800.0); p = fma(p, r, 1.0 / 3628800.0); p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0); p = fma(p, r, 1.0 / 5040.0); p = fma(p, r, 1.0 / 720.0); p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0); p = fma(p, r, 1.0 / 6.0); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
