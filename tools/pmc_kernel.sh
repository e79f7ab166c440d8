#!/bin/bash
# Development aid (GPU box): a few SQ counter sets over a short bench run for ONE kernel-name substring, per library build.
#   tools/pmc_kernel.sh <kernel substring> <lib or "cur"> [bench args]      -> per-dispatch means on stdout
K=$1; LIB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "$LIB" = cur ]; then unset TSDR_HIP_LIB; else export TSDR_HIP_LIB=$R/$LIB; fi
rm -rf $R/gpurun_out/pmck*
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmck$i -o p -- python3 $R/bench.py --quick --repeats 1 --steps 3 --warmup 1 --search-steps 1 --no-pipeline-leg $@ > /dev/null 2>&1
done
echo "== $LIB"
python3 $R/tools/pmc_db.py "$R/gpurun_out/pmck*/p_results.db" "$K" | grep -v "^=="
rm -rf $R/gpurun_out/pmck*
