#!/bin/bash
# a second build of the library next to the production one, for A/B runs in ONE gpurun call:
#   tools/build_variant.sh NAME [extra hipcc flags...]   ->  bm-nas_amd/bmnas/variants/libbmnas_NAME.so
#   BMNAS_LIB=bm-nas_amd/bmnas/variants/libbmnas_NAME.so python bench.py ...
# (variants/ is git-ignored and travels with the gpurun snapshot; delete it when the experiment is over)
set -e
cd "$(dirname "$0")/../bm-nas_amd"
name=$1; shift
obj=/tmp/bmnas_variant_$name
mkdir -p "$obj" bmnas/variants
srcs=$(python -c "from bmnas import build; print(' '.join(build._sources()))")
for s in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed "$@" \
      -c csrc/$s -o "$obj/$s.o" &
  while [ "$(jobs -r | wc -l)" -ge 7 ]; do sleep 0.2; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o bmnas/variants/libbmnas_$name.so "$obj"/*.o -ldl
ls -la bmnas/variants/libbmnas_$name.so
