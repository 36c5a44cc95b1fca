#!/bin/bash
# bench.py under several settings of one environment variable, alternating, on ONE box
# usage: tools/ab_env.sh REPS VAR value1 value2 ...   ("-" = unset)
reps=$1; var=$2; shift 2
for r in $(seq $reps); do
  for val in "$@"; do
    if [ "$val" = "-" ]; then unset $var; else export $var=$val; fi
    v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readlines()[-1])['ms_per_step']*1e3))")
    echo "$var=$val $v us"
  done
done
