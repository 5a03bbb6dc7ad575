#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel statistics of the bench command, bench lines of every workload,
# HBM traffic counters.  Everything lands under gpurun_out/r02/ ; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2 -- \
    python3 $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 --solve-steps 0 --no-dg > $O/stats_config2.json 2> $O/stats_config2.err
echo "rocprof config2 done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dg -- \
    python3 $R/bench.py --variant dg --steps 50 --warmup 5 > $O/stats_dg.json 2> $O/stats_dg.err
echo "rocprof dg done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config3 -- \
    python3 $R/bench.py --workload config3 --steps 20 --warmup 3 --cpu-steps 0 --solve-steps 0 --no-dg > $O/stats_config3.json 2> $O/stats_config3.err
echo "rocprof config3 done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config2h -- \
    python3 $R/bench.py --workload config2h --steps 30 --warmup 3 --cpu-steps 0 --solve-steps 0 --no-dg --no-overlap > $O/stats_config2h.json 2> $O/stats_config2h.err
echo "rocprof config2h done"
cd $R
python3 bench.py > $O/bench_config2.json 2> $O/bench_config2.err; echo "bench config2 done"
python3 bench.py --workload config3 --steps 20 --warmup 3 --cpu-steps 0 > $O/bench_config3.json 2> $O/bench_config3.err; echo "bench config3 done"
python3 bench.py --workload config2h --steps 30 --warmup 3 --cpu-steps 0 > $O/bench_config2h.json 2> $O/bench_config2h.err; echo "bench config2h done"
python3 bench.py --workload config5s --steps 20 --warmup 3 --cpu-steps 0 --solve-steps 0 > $O/bench_config5s.json 2> $O/bench_config5s.err; echo "bench config5s done"
python3 bench.py --variant dg --workload config3 --steps 20 --warmup 3 > $O/bench_dg_config3.json 2> $O/bench_dg_config3.err; echo "bench dg config3 done"
bash tools/collect_traffic.sh config3 cg > $O/traffic_config3.txt 2>&1; echo "traffic config3 done"
bash tools/collect_traffic.sh r1 dg > $O/traffic_dg_r1.txt 2>&1; echo "traffic dg r1 done"
