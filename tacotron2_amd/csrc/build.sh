#!/bin/bash
# Build libt2mi.so for gfx950 (MI355X).  Cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-value \
    -o ../libt2mi.so t2mi.hip "$@"
echo "built $(cd .. && pwd)/libt2mi.so"
