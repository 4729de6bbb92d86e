"""r03: device-time stamps of the stages of the SVGP training call (probe build, MXF_SVGP_STAGES=1: hipEvents on the call's three streams,
printed at the end of each call) -- the call's critical path without a profiler (whose ~30 us per launch makes a 4-sample step host-bound).
usage: svgp_stages.py [samples] [further bench.py arguments]"""
import os
import subprocess
import sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
env = dict(os.environ, MXF_SVGP_STAGES='1', MXF_GP_LIB=os.path.join(root, 'mxfusion_amd', 'libmxf_gp_probe.so'))
S = sys.argv[1] if len(sys.argv) > 1 else '4'
r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '4', '--warmup', '2', '--samples', S, '--no-cpu-baseline', '--no-extras'] + sys.argv[2:],
                   env=env, capture_output=True, text=True)
blocks = r.stderr.split('  --\n')
print(blocks[-2] if len(blocks) > 1 else r.stderr[-3000:])
