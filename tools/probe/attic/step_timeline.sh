#!/bin/bash
# kernel trace of a dozen pipelined train steps (tools/probe/trace_step.py) -> per-queue timeline of one steady step + kernel stats
# usage: bash tools/step_timeline.sh NAME [which_step]      -> gpurun_out/NAME/{step_timeline.txt,train_kernel_stats.txt}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-timeline}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/t -o s -- python3 $R/tools/probe/trace_step.py > $O/run.log 2>&1
db=$(find $O/t -name "*.db" | head -1)
python3 $R/tools/rocpd_step.py $db ${2:-12} > $O/step_timeline.txt 2>&1
python3 $R/tools/rocpd_stats.py $db 70 > $O/train_kernel_stats.txt 2>&1
rm -rf $O/t
tail -1 $O/run.log; head -3 $O/step_timeline.txt
