cd $GRAFT_REPO_ROOT
for cfg in "16:-1" "32:-1" "64:-1" "32:0" "64:0"; do
  rb=${cfg%%:*}; ka=${cfg##*:}
  echo "=== RB=$rb KA=$ka"
  MXF_SVGP_PSI2_RB=$rb MXF_SVGP_PSI2_KA=$ka python tests/probes/svgp_stages.py 32
done
