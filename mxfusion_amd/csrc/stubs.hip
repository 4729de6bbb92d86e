// Entry points declared in include/mxf_gp.h that are not implemented YET fail loudly (never silently).
#include "common.h"
#define NOTIMPL(name) MXF_FAIL(h, -99, name ": not implemented in this build")

extern "C" int mxf_gram_bwd(mxf_handle h, int, int, int, int64_t, int64_t, int, const void*, int64_t, const void*, int64_t,
                            const void*, int, int64_t, const void*, int64_t, const void*, int64_t, int64_t,
                            void*, void*, void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_gram_bwd"); }
extern "C" int mxf_potrf(mxf_handle h, int, int, int64_t, void*, int64_t, int64_t, int*, void*) { if (!h) return -1; NOTIMPL("mxf_potrf"); }
extern "C" int mxf_trsm(mxf_handle h, int, int, int, int64_t, int64_t, const void*, int64_t, int64_t, void*, int64_t, int64_t, void*) { if (!h) return -1; NOTIMPL("mxf_trsm"); }
extern "C" int mxf_trtri(mxf_handle h, int, int, int64_t, const void*, int64_t, int64_t, void*, int64_t, int64_t, void*) { if (!h) return -1; NOTIMPL("mxf_trtri"); }
extern "C" int mxf_sumlogdiag(mxf_handle h, int, int, int64_t, const void*, int64_t, int64_t, void*, void*) { if (!h) return -1; NOTIMPL("mxf_sumlogdiag"); }
extern "C" int mxf_softplus_fwd(mxf_handle h, int, int64_t, const void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_softplus_fwd"); }
extern "C" int mxf_softplus_bwd(mxf_handle h, int, int64_t, const void*, const void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_softplus_bwd"); }
extern "C" int mxf_normal_reparam(mxf_handle h, int, int, int64_t, const void*, const void*, const void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_normal_reparam"); }
extern "C" int mxf_normal_logpdf(mxf_handle h, int, int, int64_t, const void*, const void*, int64_t, const void*, int64_t, double,
                                 void*, void*, void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_normal_logpdf"); }
extern "C" int mxf_normal_reparam_bwd(mxf_handle h, int, int, int64_t, const void*, const void*, const void*, void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_normal_reparam_bwd"); }
extern "C" int mxf_adam_step(mxf_handle h, int, int64_t, void*, const void*, void*, void*, double, double, double, double, double, int, void*) { if (!h) return -1; NOTIMPL("mxf_adam_step"); }
extern "C" int mxf_gp_logpdf(mxf_handle h, int, int, int, int64_t, int, int, const void*, int64_t, const void*, int64_t, const void*, int64_t,
                             const void*, int, int64_t, const void*, int64_t, double, void*, void*, void*, int*, int,
                             void*, void*, void*, void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_gp_logpdf"); }
extern "C" int mxf_svgp_logpdf(mxf_handle h, int, int, int, int64_t, int64_t, int, int, const void*, int64_t, const void*, int64_t,
                               const void*, const void*, const void*, const void*, const void*, const void*, int, const void*,
                               double, double, double, void*, int*, int, void*, void*, void*, void*, void*, void*, void*,
                               void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_svgp_logpdf"); }
