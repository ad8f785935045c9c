#!/bin/bash
# diagnostic build of the library with in-kernel phase stamps (persist.h PZ_STAMPS); output is git-ignored
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-value -DPZ_STAMPS \
    -o tools/bin/libt2mi_stamps.so tacotron2_amd/csrc/t2mi.hip
echo built tools/bin/libt2mi_stamps.so
