#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel statistics of the bench command, bench lines of every workload.
# Everything lands under gpurun_out/$ROUND/ (ROUND=r04 by default) ; tools/refresh_profiles.py copies what is to be judged into profiles/ and
# regenerates the numbers quoted in profiles/README.md from those files.  usage: tools/collect_profiles.sh [part ...]
# parts: stats bench bench2 (default: all; each part fits one 20-minute gpurun call)
PARTS=${@:-stats bench bench2}
R=$GRAFT_REPO_ROOT
ROUND=${ROUND:-r04}
O=$R/gpurun_out/$ROUND
mkdir -p $O
for part in $PARTS; do
if [ $part = stats ]; then
  # one population of launches per file: the default line runs three legs (config 2, the 995 k-tet mesh, the DG variants) in
  # one process, so each leg gets a profiler run of its own next to the driver's full command
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_driver -- \
      python3 $R/bench.py --steps 20 --warmup 5 > $O/stats_driver.json 2> $O/stats_driver.err
  echo "rocprof: the driver's command done"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -- \
      python3 $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-config3 --no-dg > $O/stats_config2.json 2> $O/stats_config2.err
  echo "rocprof config2 (this leg only) done"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config3 -- \
      python3 $R/bench.py --workload config3 --steps 20 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg > $O/stats_config3.json 2> $O/stats_config3.err
  echo "rocprof config3 done"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2h -- \
      python3 $R/bench.py --workload config2h --steps 20 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg --no-overlap > $O/stats_config2h.json 2> $O/stats_config2h.err
  echo "rocprof config2h done"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dg -- \
      python3 $R/bench.py --variant dg --steps 50 --warmup 5 --solve-steps 0 > $O/stats_dg.json 2> $O/stats_dg.err
  echo "rocprof dg (tetrahedra, config 2) done"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dg_config2h -- \
      python3 $R/bench.py --variant dg --workload config2h --steps 20 --warmup 3 --solve-steps 0 > $O/stats_dg_config2h.json 2> $O/stats_dg_config2h.err
  echo "rocprof dg (hexahedra, config 2h, box-mesh kernels) done"
  find $O -name "*_kernel_trace.csv" -delete      # (the statistics are what is kept; the traces are 7 MB each)
fi
if [ $part = bench ]; then
  cd $R
  python3 bench.py --steps 20 --warmup 5 > $O/bench_config2.json 2> $O/bench_config2.err; echo "bench config2 done"
  python3 bench.py --workload config3 --steps 20 --warmup 5 --cpu-steps 0 --no-dg > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench config3 done"
  python3 bench.py --workload config2h --steps 20 --warmup 5 --cpu-steps 0 --no-dg > $O/bench_config2h.json 2> $O/bench_config2h.err; echo "bench config2h done"
  python3 bench.py --workload config5s --steps 20 --warmup 5 --cpu-steps 0 --no-dg > $O/bench_config5s.json 2> $O/bench_config5s.err; echo "bench config5s done"
fi
if [ $part = bench2 ]; then
  cd $R
  python3 bench.py --workload r3 --steps 10 --warmup 5 --repeats 3 --cpu-steps 0 --solve-steps 0 --no-dg > $O/bench_r3.json 2> $O/bench_r3.err; echo "bench r3 done"
  python3 bench.py --variant dg --workload config3 --steps 20 --warmup 3 > $O/bench_dg_config3.json 2> $O/bench_dg_config3.err; echo "bench dg config3 done"
  python3 bench.py --variant dg --workload config2h --steps 20 --warmup 3 --solve-steps 0 > $O/bench_dg_config2h.json 2> $O/bench_dg_config2h.err; echo "bench dg config2h (broken Q1, box-mesh kernels) done"
  KNPEMI_DG_HEX_GENERAL=1 python3 bench.py --variant dg --workload config2h --steps 20 --warmup 3 --solve-steps 0 > $O/bench_dg_config2h_general_kernels.json 2> $O/bench_dg_config2h_general.err; echo "bench dg config2h (general kernels) done"
  python3 tools/dg_solves.py --workload config2 --steps 10 --warmup 2 > $O/dg_solves_config2.json 2> $O/dg_solves_config2.err; echo "dg solves config2 done"
  KNPEMI_DG_AUX_UNSPLIT=1 KNPEMI_DG_PLAIN_AGGREGATION=1 KNPEMI_DG_AUX_SMOOTHED=1 python3 tools/dg_solves.py --workload config2 --steps 10 --warmup 2 > $O/dg_solves_config2_continuous_aux_space.json 2> $O/dg_solves_config2_cont.err; echo "dg solves config2 (round-2 auxiliary space) done"
  python3 tools/dg_solves.py --workload hex_r1 --steps 10 --warmup 2 > $O/dg_solves_hex_r1.json 2> $O/dg_solves_hex_r1.err; echo "dg solves hexahedra r=1 done"
  python3 tools/dg_solves.py --workload config2h --steps 10 --warmup 2 --solve-steps 5 > $O/dg_solves_config2h.json 2> $O/dg_solves_config2h.err; echo "dg solves config2h done"
  KNPEMI_NO_FUSED=1 python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-dg --no-config3 > $O/bench_config2_plain_solver_loops.json 2> $O/bench_config2_plain.err; echo "bench config2 (plain solver loops) done"
fi
done
