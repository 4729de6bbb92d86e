#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05_b2
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_training_f32.py > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
python -m pytest tests/test_gpu_training_f32.py -q -x -s > $O/pytest_train.log 2>&1; tail -15 $O/pytest_train.log
for a in "--minibatch 8192 --shard rows --samples 1" "--minibatch 8192 --shard rows --samples 1 --proxy-world 8" "--minibatch 8192 --samples 4" "--minibatch 8192 --samples 32"; do
  n=$(echo $a | tr -d ' -')
  python bench.py $a --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_$n.json 2> $O/bench_$n.err; cat $O/bench_$n.json; tail -3 $O/bench_$n.err
done
export MXF_GP_LIB=$R/mxfusion_amd/libmxf_gp_probe.so
cd /tmp && export TMPDIR=/tmp
MXF_SPLIT_PAIR=1 MXF_SPLIT_SYNC=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_v1ns_FETCH -o g -- python $R/tests/probes/split_pmc.py v > $O/pmc_v1ns.log 2>&1
python $R/profiles/pmc_summary.py gemm_f16x2 $O/gemm_v1ns_pmc.json $O/pmc_v1ns_FETCH > $O/gemm_v1ns_pmc.txt 2>&1; cat $O/gemm_v1ns_pmc.txt
