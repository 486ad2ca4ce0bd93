#!/bin/bash
# Dev helper: build the library, stop on a failed build, then run a command on the GPU box.   usage: gpu.sh [--timeout S] 'command'
set -e
T=900
if [ "$1" == "--timeout" ]; then T=$2; shift 2; fi
cd "$(dirname "$0")/.."
python ssd-object-detection_amd/build.py > /tmp/build.log 2>&1 || { grep -E "error" -A4 /tmp/build.log | head -30; echo "BUILD FAILED"; exit 1; }
/usr/local/graft/bin/gpurun --timeout $T -- "$1"
