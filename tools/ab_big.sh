for b in 256 512 1024; do
  for lib in prod base; do
    L="X=1"; [ $lib != prod ] && L="BMNAS_LIB=bm-nas_amd/bmnas/variants/libbmnas_$lib.so"
    env $L python bench.py --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-full-step --no-roofline > /tmp/x.json 2>/tmp/x.err
    python -c "import json; d=json.loads(open('/tmp/x.json').read().strip().splitlines()[-1]); print('b$b $lib', d['ms_per_step'])"
  done
done
