#!/bin/bash
# Round profile set of the default bench command (run on the GPU box): kernel trace + stats, then
# PMC passes (counters alone, no trace domains), the wave timeline and the launch-rate probe.
# usage: bash tools/collect_profiles.sh <tag>     -> gpurun_out/profiles_<tag>/
set -e
tag=$1
root=$PWD
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --no-also --steps 300 --warmup 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p -- $B > $out/p$i.log 2>&1
done
python3 $root/profiles/summarize_pmc.py $out/p*/*counter_collection.csv > $out/pmc.json
python3 $root/tools/traffic_from_pmc.py $out/pmc.json metric 4096 10 "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_profiles.sh $tag), bench.py --steps 300" > $out/traffic_metric.json
cd $root
python3 bench.py --steps 500 --warmup 20 > $out/bench_default.jsonl 2>$out/bench_default.err
if [ -f eb-cadrl_amd/lib/libebcsim_trace.so ]; then
  EBCSIM_LIB=$root/eb-cadrl_amd/lib/libebcsim_trace.so python3 tools/wave_timeline.py metric > $out/wave_timeline.txt 2>&1
fi
[ -x tools/bin/launch_rate ] && tools/bin/launch_rate > $out/launch_rate.txt 2>&1
# the robot decision (look-ahead sweep + value network): per-kernel times
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/sarl -o sarl -- python3 $root/tools/sarl_profile.py 1024 > $out/sarl.log 2>&1) || true
ls $out
