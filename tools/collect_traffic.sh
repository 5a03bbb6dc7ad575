#!/bin/bash
# HBM traffic of the hot kernels from the L2's memory-side counters, as /opt/skills/guides/MI355X_MICROARCH.md (HBM
# section) prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass, so each gets its own rocprofv3 run with
# --kernel-trace only; the per-launch averages go to profiles/rNN_traffic.json (tools/traffic_summary.py applies the
# gfx950 correction when bench.py reads them: bytes = 2 * FETCH_SIZE + WRITE_SIZE, KiB units).
# usage (on the GPU box): tools/collect_traffic.sh <workload> [cg | dg [tetrahedron|hexahedron [general]]]
#   cg: bench.py --workload <workload>;  dg: tools/dg_time.py -r <workload's refinement, e.g. r2> --cell <cell>
set -e
W=${1:-config2}; MODE=${2:-cg}; CELL=${3:-tetrahedron}; GEN=${4:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${MODE}
if [ "$MODE" = dg ]; then TAG=dg_${CELL}${GEN:+_general}; fi
OUT=$R/gpurun_out/traffic_${W}_${TAG}
if [ -n "$GEN" ]; then export KNPEMI_DG_HEX_GENERAL=1; fi
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  if [ "$MODE" = dg ]; then
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- \
        python3 $R/tools/dg_time.py -r ${W#r} --cell $CELL --reps 3 > $OUT.$c.log 2>&1 || echo "pass $c failed"
  else
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- \
        python3 $R/bench.py --workload $W --steps 5 --warmup 2 --repeats 1 --cpu-steps 0 --solve-steps 0 --no-dg --no-config3 > $OUT.$c.log 2>&1 || echo "pass $c failed"
  fi
  echo "pass $c done"
done
python3 $R/tools/traffic_summary.py $OUT $W $TAG
# (the per-launch averages are in $OUT.json; the raw passes are tens of MB)
rm -rf $OUT
