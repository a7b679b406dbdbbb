#!/bin/bash
# A/B of two builds of the library on one box: tools/ab_brax.sh [variant]   (gpurun_in/libcarl_<variant>.so, default "old")
# (1) do they compute the same?  digests + the largest difference of the final states   (2) launch times, three families
export CARL_AMD_NO_BUILD=1
V=${1:-old}
mkdir -p gpurun_out
python tools/brax_digest.py --dump /tmp/d_new.npz > gpurun_out/digest_new.txt 2>gpurun_out/digest_new.err
CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$V.so python tools/brax_digest.py --dump /tmp/d_old.npz > gpurun_out/digest_$V.txt 2>gpurun_out/digest_$V.err
diff gpurun_out/digest_$V.txt gpurun_out/digest_new.txt > /dev/null && echo DIGESTS IDENTICAL
python tools/brax_digest.py --compare /tmp/d_old.npz /tmp/d_new.npz
[ -n "$NO_BENCH" ] && exit 0
for e in ${ENVS:-ant halfcheetah humanoid}; do
for rep in 1 2; do for v in base $V; do
  if [ "$v" = base ]; then unset CARL_AMD_LIB_PATH; else export CARL_AMD_LIB_PATH=$PWD/gpurun_in/libcarl_$v.so; fi
  python bench.py --env $e --lanes 32768 --steps 40 --warmup 5 --no-cpu-baseline --no-per-call --also none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$e $v', 'launch_ms %.3f value %.3e'%(r['avg_launch_ms'], d['value']))"
done; done; done
