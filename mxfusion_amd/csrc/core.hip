// Handle management for libmxf_gp.so.
#include "common.h"
#include "internal.h"

extern "C" int mxf_version(void) { return 100; }

extern "C" int mxf_create(int device, mxf_handle* out) {
    if (!out) return -1;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return -2;
    if (hipSetDevice(device) != hipSuccess) return -3;
    mxf_ctx* h = new mxf_ctx();
    h->device = device;
    *out = h;
    return 0;
}

extern "C" int mxf_destroy(mxf_handle h) {
    if (!h) return -1;
    mxf_comm_release(h);
    if (h->ws) (void)hipFree(h->ws);
    if (h->gram_ws) (void)hipFree(h->gram_ws);
    if (h->flags) (void)hipFree(h->flags);
    if (h->gsync) (void)hipFree(h->gsync);
    if (h->pinv) (void)hipFree(h->pinv);
    if (h->tm.made) for (int i = 0; i < 2 * MXF_NT; ++i) (void)hipEventDestroy(h->tm.ev[i]);
    if (h->cond_dev) (void)hipFree(h->cond_dev);
    if (h->cond_host) (void)hipHostFree(h->cond_host);
    if (h->bwd_acc) (void)hipFree(h->bwd_acc);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
    if (h->ev_aux) (void)hipEventDestroy(h->ev_aux);
    if (h->ev_aux2) (void)hipEventDestroy(h->ev_aux2);
    if (h->ev_tg) (void)hipEventDestroy(h->ev_tg);
    if (h->ev_su) (void)hipEventDestroy(h->ev_su);
    if (h->ev_k1) (void)hipEventDestroy(h->ev_k1);
    if (h->ev_k2) (void)hipEventDestroy(h->ev_k2);
    if (h->ev_k3) (void)hipEventDestroy(h->ev_k3);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->side2) (void)hipStreamDestroy(h->side2);
    if (h->potrf_aux) (void)hipStreamDestroy(h->potrf_aux);
    if (h->ev_pa) (void)hipEventDestroy(h->ev_pa);
    if (h->ev_pb) (void)hipEventDestroy(h->ev_pb);
    if (h->ev_ph) (void)hipEventDestroy(h->ev_ph);
    if (h->potrf_inv) (void)hipStreamDestroy(h->potrf_inv);
    if (h->ev_pi) (void)hipEventDestroy(h->ev_pi);
    if (h->ev_pj) (void)hipEventDestroy(h->ev_pj);
    if (h->potrf_rows) (void)hipStreamDestroy(h->potrf_rows);
    if (h->ev_pc) (void)hipEventDestroy(h->ev_pc);
    if (h->ev_rb) (void)hipEventDestroy(h->ev_rb);
    if (h->potrf_acc) (void)hipStreamDestroy(h->potrf_acc);
    if (h->ev_pq) (void)hipEventDestroy(h->ev_pq);
    if (h->ev_pz) (void)hipEventDestroy(h->ev_pz);
    if (h->potrf_chain) (void)hipStreamDestroy(h->potrf_chain);
    if (h->ev_pk) (void)hipEventDestroy(h->ev_pk);
    delete h;
    return 0;
}

extern "C" const char* mxf_last_error(mxf_handle h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int64_t mxf_workspace_bytes(mxf_handle h) { return h ? (int64_t)(h->ws_bytes + h->gram_ws_bytes + h->bwd_acc_bytes + h->pinv_elems * sizeof(double)) : -1; }

extern "C" int64_t mxf_workspace_generation(mxf_handle h) { return h ? h->ws_generation : -1; }

extern "C" int mxf_svgp_cond_nowait(mxf_handle h, double* cond1_max_out, int reset) {
    if (!h || !cond1_max_out) return -1;
    double m = 0.0;      // no synchronisation: whatever the finished calls have published, over every slot
    if (h->cond_host)
        for (int i = 0; i < MXF_COND_SLOTS; ++i) {
            const double v = *(volatile double*)(h->cond_host + 2 * i);
            m = v > m ? v : m;
            if (reset) { *(volatile double*)(h->cond_host + 2 * i) = 0.0; *(volatile double*)(h->cond_host + 2 * i + 1) = 0.0; }
        }
    *cond1_max_out = m;
    return 0;
}

extern "C" int mxf_svgp_configure(mxf_handle h, int form, int cond_slot) {
    if (!h) return -1;
    if (form != MXF_SVGP_EXPLICIT && form != MXF_SVGP_WHITENED) MXF_FAIL(h, -2, "mxf_svgp_configure: unknown form %d", form);
    if (cond_slot < 0 || cond_slot >= MXF_COND_SLOTS) MXF_FAIL(h, -2, "mxf_svgp_configure: slot %d outside [0, %d)", cond_slot, MXF_COND_SLOTS);
    h->svgp_form = form; h->cond_slot = cond_slot;
    return 0;
}

extern "C" int mxf_svgp_cond_slot(mxf_handle h, int slot, double* last_out, double* max_out, int reset) {
    if (!h) return -1;
    if (slot < 0 || slot >= MXF_COND_SLOTS) MXF_FAIL(h, -2, "mxf_svgp_cond_slot: slot %d outside [0, %d)", slot, MXF_COND_SLOTS);
    if (max_out) *max_out = h->cond_host ? *(volatile double*)(h->cond_host + 2 * slot) : 0.0;
    if (last_out) *last_out = h->cond_host ? *(volatile double*)(h->cond_host + 2 * slot + 1) : 0.0;
    if (reset && h->cond_host) { *(volatile double*)(h->cond_host + 2 * slot) = 0.0; *(volatile double*)(h->cond_host + 2 * slot + 1) = 0.0; }
    return 0;
}

extern "C" int mxf_svgp_timing(mxf_handle h, int enable) {
    if (!h) return -1;
    if (enable && !h->tm.init()) MXF_FAIL(h, -4, "mxf_svgp_timing: cannot create events");
    h->tm.on = enable != 0;
    for (int i = 0; i < MXF_NT; ++i) h->tm.used[i] = false;
    return 0;
}

extern "C" int mxf_svgp_timing_read(mxf_handle h, double* ms_out) {
    if (!h || !ms_out) return -1;
    for (int i = 0; i < MXF_NT; ++i) ms_out[i] = -1.0;
    if (!h->tm.made) return 0;
    MXF_HIP(h, hipDeviceSynchronize());
    for (int i = 0; i < MXF_NT; ++i) {
        if (!h->tm.used[i]) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->tm.ev[2 * i], h->tm.ev[2 * i + 1]) == hipSuccess) ms_out[i] = (double)ms;
    }
    return 0;
}

extern "C" int mxf_svgp_whitened_ok(int dtype, int S, int64_t B, int64_t M, int Q, int P, int64_t strideS_X) {
    const int64_t SB = (strideS_X == 0 ? 1 : (int64_t)S) * B;
    return (dtype == MXF_F32 && S > 0 && B > 0 && (M % 128) == 0 && M >= 128 && (SB % 256) == 0 && Q <= 16 && P <= 8 && !(S > 1 && strideS_X == 0)) ? 1 : 0;
}

extern "C" int mxf_svgp_last_cond(mxf_handle h, double* cond1_out) {
    if (!h || !cond1_out) return -1;
    *cond1_out = 0.0;
    if (!h->cond_dev) return 0;                      // no SVGP training call on this handle yet
    double v[2];
    // the norms are written by kernels on the caller's (possibly non-blocking) stream: a plain hipMemcpy orders against the null stream only
    MXF_HIP(h, hipDeviceSynchronize());
    MXF_HIP(h, hipMemcpy(v, h->cond_dev + 2, sizeof(v), hipMemcpyDeviceToHost));     // (words 2, 3: the finished call's norms; 0, 1 are the accumulators of the next)
    *cond1_out = v[0] * v[1];
    return 0;
}
