#!/bin/bash
# PMC passes over tools/profile_variants.py (counters only, no trace domains), one pass per group.
# usage (on the GPU box): bash tools/pmc_variants.sh <outdir> [env assignments...]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -o p -- python3 /root/repo/tools/profile_variants.py metric > $out/p$i.log 2>&1
done
python3 /root/repo/profiles/summarize_pmc.py $out/p*/*counter_collection.csv > $out/pmc.json
