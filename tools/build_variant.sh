#!/bin/bash
# Builds an experiment variant of the library into tools/alt/<name>.so (in-tree so that it travels with gpurun; the
# directory is git-ignored):   tools/build_variant.sh <name> [-DFLAG ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/tools/alt"
cd "$ROOT/eaqhm-analysis-and-synthesis-in-python_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fno-strict-aliasing --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-result "$@" \
  -o "$ROOT/tools/alt/$NAME.so" eaqhm_api.hip eaqhm_ls.hip eaqhm_ls_mfma.hip eaqhm_ls_tile.hip eaqhm_interp.hip
echo "built tools/alt/$NAME.so"
