#!/bin/bash
export CARL_AMD_NO_BUILD=1 TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | grep real
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
