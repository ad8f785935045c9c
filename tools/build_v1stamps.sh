#!/bin/bash
# diagnostic build of the library with stage stamps in the V1 trainer's persistent recurrences (v1_dec_train.h V1D_STAMPS); output is git-ignored
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-value -DV1D_STAMPS \
    -o tools/bin/libt2mi_v1stamps.so tacotron2_amd/csrc/t2mi.hip
echo built tools/bin/libt2mi_v1stamps.so
