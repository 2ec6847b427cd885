#!/bin/bash
# kernel-trace of tools/serial_step.py -> gpurun_out/serial/serial_kernel_stats.txt (per-kernel alone-time table of one train step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/serial; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/serial_step.py > $O/untraced.txt 2>&1
timeout 600 rocprofv3 --kernel-trace -d $O/t -o s -- python3 $R/tools/serial_step.py > $O/traced.txt 2>&1
db=$(find $O/t -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db 80 > $O/serial_kernel_stats.txt 2>&1
rm -rf $O/t
cat $O/untraced.txt; tail -2 $O/traced.txt; head -45 $O/serial_kernel_stats.txt | cut -c1-70,100-190
