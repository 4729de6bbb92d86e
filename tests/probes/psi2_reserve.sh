# step time against the CUs the Psi2 product leaves free for the core chains and the second planes pass (probe-build knobs
# MXF_SVGP_PSI2_RB = CUs reserved in phase B, MXF_SVGP_PSI2_KA = columns of the reduced-occupancy phase A; -1 = default)
cd $GRAFT_REPO_ROOT
export MXF_GP_LIB=$PWD/mxfusion_amd/libmxf_gp_probe.so
for S in ${1:-32}; do for cfg in ${2:-"16:-1 32:-1 48:-1 64:-1 80:-1 48:0 64:0"}; do
  rb=${cfg%%:*}; ka=${cfg##*:}
  echo -n "S=$S RB=$rb KA=$ka  "
  MXF_SVGP_PSI2_RB=$rb MXF_SVGP_PSI2_KA=$ka python bench.py --steps 20 --warmup 4 --samples $S --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
done; done
