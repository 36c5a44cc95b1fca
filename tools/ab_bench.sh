#!/bin/bash
# A/B on ONE box: alternate bench.py between library builds (paths after the first argument = reps)
reps=$1; shift
for r in $(seq $reps); do
  for lib in "$@"; do
    v=$(EBCSIM_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readlines()[-1])['ms_per_step']*1e3))")
    echo "$(basename $lib) $v us"
  done
done
