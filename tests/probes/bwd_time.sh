# kernel times of the SVGP reverse pass and of the split GEMMs inside the bench step (rocprofv3 kernel trace of 3 steps); optional MXF_GP_LIB
# selects an experiment build
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/bwdtime
rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
grep "svgp_bwd_mfma\|gram_bwd_kernel<float\|gemm_f16x2_wide\|gram_planes\|bwd_prescale" $(find $O -name '*kernel_stats.csv') | cut -d, -f1-5 | cut -c1-150
