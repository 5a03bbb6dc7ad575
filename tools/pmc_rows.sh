#!/bin/bash
# PMC passes for the row kernels (one counter group per rocprofv3 run, kernel trace only).
# usage: tools/pmc_rows.sh <workload> <outdir under gpurun_out>
set -e
W=${1:-r3}; OUT=${2:-pmc_rows}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/$OUT/p$i -- \
      python3 $R/bench.py --workload $W --steps 3 --warmup 1 --repeats 1 --cpu-steps 0 --solve-steps 0 --no-overlap --no-dg --no-config3 > $R/gpurun_out/$OUT.p$i.log 2>&1
  echo "pass $i ($grp) done"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/$OUT
# the summary is what is kept: the per-pass traces are tens of MB and gpurun merges at most 64 MiB back
rm -rf $R/gpurun_out/$OUT/p[0-9]*
