# probe build of the library with stage timestamps inside potrf_tiles_kernel (for potrf_trace.py): mxfusion_amd/libmxf_gp_trace.so
cd "$(dirname "$0")/../../mxfusion_amd/csrc" && make -j8 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics -DMXF_POTRF_TRACE -c chol.hip -o /tmp/chol_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v '^chol.o$') /tmp/chol_trace.o -o ../libmxf_gp_trace.so
