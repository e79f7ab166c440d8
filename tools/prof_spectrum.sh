#!/bin/bash
# Development aid (GPU box): kernel times and SQ counters of the spectrum legs (tools/time_spectrum.py).
TAG=${1:-sp}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/${TAG}_sp_stats $O/${TAG}_sp_pmc*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_sp_stats -o run -- python3 $R/tools/time_spectrum.py > /dev/null 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/${TAG}_sp_pmc$i -o p -- python3 $R/tools/time_spectrum.py > /dev/null 2>&1
done
python3 $R/tools/pmc_db.py "$O/${TAG}_sp_pmc*/p_results.db" k_seg1024 k_welch > $O/${TAG}_sp_pmc.txt 2>&1
head -12 $O/${TAG}_sp_stats/run_kernel_stats.csv | cut -c1-160
