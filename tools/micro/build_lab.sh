#!/bin/bash
# Build the stand-alone smoothing labs into tools/micro/bin (git-ignored; travels to the GPU box with gpurun).
#   tools/micro/build_lab.sh NAME [extra -D flags]      e.g.  build_lab.sh fast -DSMOOTH_FAST
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/bin"
name="$1"; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -DSMOOTH_SRC="\"$ROOT/meshdqn_amd/csrc/mdq_smooth.hip\"" \
  "$HERE/smooth_lab.hip" -o "$HERE/bin/smooth_lab_$name"
echo "built $HERE/bin/smooth_lab_$name"
