#!/bin/bash
# PMC passes for the DG assembly kernels (one counter group per rocprofv3 run, kernel trace only).
# usage: tools/pmc_dg.sh <refinement r> <outdir under gpurun_out>
set -e
RR=${1:-1}; OUT=${2:-pmc_dg}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
           "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/$OUT/p$i -- \
      python3 $R/tools/dg_time.py -r $RR --reps 3 $DG_TIME_ARGS > $R/gpurun_out/$OUT.p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i ($grp) done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/$OUT
# the summary is what is kept: the per-pass traces are tens of MB and gpurun merges at most 64 MiB back
rm -rf $R/gpurun_out/$OUT/p[0-9]*
