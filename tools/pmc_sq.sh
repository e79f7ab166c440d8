#!/bin/bash
# Development aid (GPU box): SQ / GRBM counter passes over a short bench run; rocpd databases land in
# gpurun_out/pmc_sq*/ and tools/pmc_db.py prints per-kernel means.  Extra arguments go to bench.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sq*
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_sq$i -o p -- python3 $R/bench.py --quick --repeats 1 --steps 3 --warmup 1 --search-steps 1 $@ > /dev/null 2>&1
done
python3 $R/tools/pmc_db.py "$R/gpurun_out/pmc_sq*/p_results.db" k_proj k_beta k_fold k_shift k_raster
