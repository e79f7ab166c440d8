#!/bin/bash
# Development aid (GPU box): three SQ counter sets over ANY python command, per-dispatch means for one kernel-name substring.
#   tools/pmc_cmd.sh <kernel substring> <script.py> [args]        e.g.  tools/pmc_cmd.sh k_seg bench.py --spectra-only welch
K=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=$R/$1; shift
rm -rf $R/gpurun_out/pmcc*
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcc$i -o p -- python3 $S "$@" > /dev/null 2>&1
done
python3 $R/tools/pmc_db.py "$R/gpurun_out/pmcc*/p_results.db" "$K"
rm -rf $R/gpurun_out/pmcc*
