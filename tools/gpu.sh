#!/bin/bash
# build the gfx950 library here (the snapshot carries the .so), then run a command on the GPU box
#   tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
(cd bm-nas_amd && python -m bmnas.build > /dev/null)
exec /usr/local/graft/bin/gpurun "$@"
