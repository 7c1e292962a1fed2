cd /root/repo
timeout 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in 1 0 1 0; do
BMNAS_AB=$v timeout 200 python - <<PY
import os, sys, json, subprocess
sys.path.insert(0, 'bm-nas_amd')
import bmnas.cell as K
K.FUSE_ATTN_GEMM = bool(int(os.environ['BMNAS_AB']))
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline', '--no-full-step', '--steps', '300']
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit:
    pass
PY
done 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms_per_step', d['ms_per_step'])
    elif 'passed' in l or 'failed' in l: print(l.strip())
"
