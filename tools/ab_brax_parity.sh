#!/bin/bash
export CARL_AMD_NO_BUILD=1
for v in base "$@"; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  echo "== $v"; python tools/diag_brax_parity.py 2>&1 | grep n_frames
done
