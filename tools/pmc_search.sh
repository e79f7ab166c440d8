#!/bin/bash
# Development aid (GPU box): SQ counter sets over the configuration search (tools/prof_search.py) for one kernel-name substring.
#   tools/pmc_search.sh <kernel substring> [n]
K=$1; N=${2:-4000000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcs*
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_INSTS_VALU_FMA_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcs$i -o p -- python3 $R/tools/prof_search.py $N search > /dev/null 2>&1
done
python3 $R/tools/pmc_db.py "$R/gpurun_out/pmcs*/p_results.db" "$K"
rm -rf $R/gpurun_out/pmcs*
