// Entry points declared in include/mxf_gp.h that are not implemented YET fail loudly (never silently).
#include "common.h"
#define NOTIMPL(name) MXF_FAIL(h, -99, name ": not implemented in this build")

extern "C" int mxf_gp_logpdf(mxf_handle h, int, int, int, int64_t, int, int, const void*, int64_t, const void*, int64_t, const void*, int64_t,
                             const void*, int, int64_t, const void*, int64_t, double, void*, void*, void*, int*, int,
                             void*, void*, void*, void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_gp_logpdf"); }
extern "C" int mxf_svgp_logpdf(mxf_handle h, int, int, int, int64_t, int64_t, int, int, const void*, int64_t, const void*, int64_t,
                               const void*, const void*, const void*, const void*, const void*, const void*, int, const void*,
                               double, double, double, void*, int*, int, void*, void*, void*, void*, void*, void*, void*,
                               void*, void*, void*) { if (!h) return -1; NOTIMPL("mxf_svgp_logpdf"); }
