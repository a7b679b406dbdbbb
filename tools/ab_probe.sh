#!/bin/bash
# A/B of the product library against gpurun_in/libcarl_<V>.so on ONE box, interleaved: tools/ab_probe.sh "<probe cases>" V...
# (cases = names of tools/shard8_probe.py, comma separated)
export CARL_AMD_NO_BUILD=1
cases=$1; shift
for rep in 1 2 3; do
for v in base "$@"; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  python tools/shard8_probe.py --reps 3 --only $cases 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('rep $rep %-6s %-28s %8.1f us  (min %.1f)'%('$v', d['name'], d['event_us'], d['min_us']))"
done; done
