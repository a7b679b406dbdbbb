#!/bin/bash
# build (cross-compiles anywhere) and run (GPU box) the MFMA-vs-VALU inertia contraction experiment
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 inertia_bench.hip -o inertia_bench
./inertia_bench ${1:-200}
