# probe build of the library with stage stamps inside svgp_bwd_mfma_kernel (for bwd_trace.py): mxfusion_amd/libmxf_gp_bwdtrace.so
cd "$(dirname "$0")/../../mxfusion_amd/csrc" && make -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -w -DMXF_BWD_TRACE -c gram_bwd.hip -o /tmp/gram_bwd_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^gram_bwd.o$') /tmp/gram_bwd_trace.o -o ../libmxf_gp_bwdtrace.so
