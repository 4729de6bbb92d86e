cd $GRAFT_REPO_ROOT; O=gpurun_out/sec_r05; mkdir -p $O
python bench.py > $O/r05_bench_full.json 2> $O/full.log
python bench.py --trained-like --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/r05_bench_trained_like.json 2>/dev/null
python bench.py --samples 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r05_bench_4samples_per_gpu.json 2>/dev/null
python bench.py --samples 4 --trained-like --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r05_bench_4samples_per_gpu_trained_like.json 2>/dev/null
python bench.py --samples 4 --steps 30 --warmup 5 --no-cpu-baseline --no-extras --force-dist > $O/r05_bench_4samples_per_gpu_rccl_path.json 2>/dev/null
python bench.py --minibatch 8192 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_minibatch8192.json 2>/dev/null
python bench.py --minibatch 8192 --samples 4 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_minibatch8192_4samples.json 2>/dev/null
python bench.py --minibatch 8192 --shard rows --samples 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_rows_minibatch8192_1gpu.json 2>/dev/null
python bench.py --minibatch 8192 --shard rows --samples 1 --proxy-world 8 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_rows_minibatch8192_rank_of_8.json 2>/dev/null
python bench.py --minibatch 65536 --shard rows --samples 1 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_rows_fullbatch_1gpu.json 2>/dev/null
python bench.py --minibatch 65536 --shard rows --samples 1 --proxy-world 8 --steps 40 --warmup 5 --no-cpu-baseline > $O/r05_bench_rows_fullbatch_rank_of_8.json 2>/dev/null
python bench.py --workload gp --dtype float64 --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_exact_gp_f64.json 2>/dev/null
python bench.py --workload deepgp --samples 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r05_bench_deepgp_4samples.json 2>/dev/null
python bench.py --workload deepgp --samples 32 --steps 5 --warmup 2 --no-cpu-baseline > $O/r05_bench_deepgp_32samples.json 2>/dev/null
python bench.py --workload pilco --dtype float64 --graph 1 --steps 5 --warmup 3 --no-cpu-baseline > $O/r05_bench_pilco_f64_graph.json 2>/dev/null
for f in $O/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],3), d.get('ms_per_step_through_run'), d.get('ms_per_step_trained_like'), d.get('roofline',{}).get('frac'), d.get('roofline_mfma',{}).get('frac'), d.get('roofline_mfma_psi2',{}).get('frac'), d.get('roofline_f64_mfma',{}).get('frac'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
