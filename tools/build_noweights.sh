#!/bin/bash
# diagnostic build: the persistent decoder without weight traffic (persist.h PZ_NO_WEIGHT_STREAM); results are wrong on purpose,
# only the step time means anything (upper bound of what LDS / register residency of the weights could save); git-ignored output
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-value -DPZ_NO_WEIGHT_STREAM \
    -o tools/bin/libt2mi_noweights.so tacotron2_amd/csrc/t2mi.hip
echo built tools/bin/libt2mi_noweights.so
