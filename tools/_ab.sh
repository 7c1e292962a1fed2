set -e
Q="--no-cpu-baseline --no-full-step --no-roofline --steps 300"
for b in 32 64 256 1024; do for o in 0 2; do
BMNAS_BWD_ORDER=$o python bench.py $Q --batch $b > gpurun_out/ab_o.json 2>gpurun_out/ab_o.err
python - $b $o <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab_o.json').read().strip().splitlines()[-1])
print(f'batch {sys.argv[1]} order {sys.argv[2]}: step {d["ms_per_step"]}')
PY
done; done
python -m pytest tests -m gpu -q -x > gpurun_out/full_suite.log 2>&1; tail -3 gpurun_out/full_suite.log
