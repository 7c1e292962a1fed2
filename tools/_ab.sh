set -e
Q="--no-cpu-baseline --no-full-step --steps 400"
for v in base nola2 base nola2; do
if [ $v = base ]; then unset BMNAS_LIB; else export BMNAS_LIB=$PWD/bm-nas_amd/bmnas/variants/libbmnas_$v.so; fi
python bench.py $Q > gpurun_out/ab_$v.json 2>gpurun_out/ab_$v.err
python - $v <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/ab_{sys.argv[1]}.json').read().strip().splitlines()[-1])
r={x['kernel']:x['avg_us'] for x in d['roofline_kernels']}
print(f'{sys.argv[1]}: step {d["ms_per_step"]}  fwd {r["conv_pipe_fwd_sdpa_k<32, 3, 2>"]} bwd_all {r["conv_bwd_all_pipe_k<48, 3, 2>"]}')
PY
done
unset BMNAS_LIB
python -m pytest tests/test_kernels_gpu.py tests/test_network_gpu.py tests/test_reshape_group_gpu.py -q -x 2>&1 | tail -3
