#!/bin/bash
export CARL_AMD_NO_BUILD=1
mkdir -p gpurun_out/r02f
echo "== base (fp32 joint geometry)"; python tools/brax_parity_percentiles.py 2>&1 | grep -v Warn | tee gpurun_out/r02f/parity_base.txt
echo "== F64REL"; CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_BX_F64REL.so python tools/brax_parity_percentiles.py 2>&1 | grep -v Warn | tee gpurun_out/r02f/parity_f64rel.txt
