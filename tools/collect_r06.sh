#!/bin/bash
# Round-6 evidence in one GPU call: kernel traces (pipelined bench run, serial step), whole-step HBM traffic, per-level counter passes
# (tools/pmc_pieces.sh), the FPS counter passes, the two bench lines.  Everything lands under gpurun_out/$1; the summaries are copied
# into profiles/ afterwards (tools/pmc_pieces_summary.py r06, tools/make_profile_summaries.py <dir> r06).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06final}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VOTENET_BENCH_NO_DP_LEG=1
timeout 900 rocprofv3 --kernel-trace -d $O/train -o t -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --headline-only > $O/train_trace.log 2>&1
python3 $R/tools/rocpd_stats.py $(find $O/train -name "*.db" | head -1) 70 > $O/train_kernel_stats.txt 2>&1; rm -rf $O/train
unset VOTENET_BENCH_NO_DP_LEG
cd $R
bash tools/serial_step.sh > $O/serial.log 2>&1
bash tools/serial_last_step.sh > $O/serial_last.log 2>&1
cp gpurun_out/serial/serial_kernel_stats.txt $O/serial_kernel_stats.txt; cp gpurun_out/serial/last_step.txt $O/serial_last_step.txt; cp gpurun_out/serial/untraced.txt $O/serial_untraced.txt
bash tools/step_traffic.sh $(basename $O)_traffic > $O/traffic.log 2>&1; cp gpurun_out/$(basename $O)_traffic/step_traffic.txt $O/step_traffic.txt
bash tools/pmc_pieces.sh > $O/pmc_pieces.log 2>&1
bash tools/pmc_fps.sh $(basename $O)_pmcfps > $O/pmc_fps.log 2>&1
python3 tools/probe/stretch_time.py > $O/stretch_time.txt 2>&1
python3 tools/cpu_issue_time.py > $O/issue_pinned.txt 2>&1
python3 bench.py > $O/train_bench_line.json 2> $O/train_bench.err
python3 bench.py --workload fwd > $O/fwd_bench_line.json 2> $O/fwd_bench.err
tail -c 300 $O/train_bench_line.json; head -3 $O/stretch_time.txt; tail -3 $O/issue_pinned.txt
