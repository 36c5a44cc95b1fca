cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/vn_pmc
mkdir -p $O
python3 $R/tools/mlp2_pmc.py > $O/plain.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]*(MFMA|LDS|VALU|WAIT|BUSY|ACTIVE)[A-Z_0-9]*" | sort -u > $O/counters.txt
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o p -- python3 $R/tools/mlp2_pmc.py 524288 3 > $O/p$i.log 2>&1
done
python3 $R/profiles/summarize_pmc.py $O/p*/*counter_collection.csv > $O/pmc.json 2>$O/sum.err
cat $O/plain.txt
