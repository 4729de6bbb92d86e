# same-box A/B of two builds of the library: step time of the bench (32 and 4 samples), alternating, three rounds
# usage: ab_step.sh <libA.so> <libB.so>
for round in 1 2 3; do for lib in "$@"; do
  for S in 32 4; do
    ms=$(MXF_GP_LIB=$PWD/$lib python bench.py --steps $([ $S = 32 ] && echo 12 || echo 40) --warmup 4 --samples $S --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.readlines()[-1])['ms_per_step'],3))")
    echo "round $round $lib S=$S $ms"
  done
done; done
