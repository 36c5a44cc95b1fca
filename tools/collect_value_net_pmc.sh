#!/bin/bash
# Matrix-pipe occupancy of the value network's blocks in a decision (run on the GPU box): counters alone, one group
# per pass, over tools/sarl_profile.py.   usage: bash tools/collect_value_net_pmc.sh <tag>  -> gpurun_out/vn_pmc_<tag>/
set -e
tag=$1
root=$PWD
out=$root/gpurun_out/vn_pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export EBCSIM_CHUNK_STREAMS=1  # one stream: per-kernel times and counters without overlap
B="python3 $root/tools/sarl_profile.py 1024"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B > $out/kt.log 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p -- $B > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/failed.txt
done
python3 $root/profiles/summarize_pmc.py $out/p*/*counter_collection.csv > $out/pmc.json
cd $root
python3 tools/value_net_pmc_table.py $out/pmc.json $out/kt $out/value_net_mfma_busy.json > $out/table.txt
cat $out/table.txt
