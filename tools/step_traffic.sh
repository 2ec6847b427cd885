#!/bin/bash
# HBM traffic of the whole train step by kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md
# prescribes) over a short bench run -> gpurun_out/$1/step_traffic.txt (tools/step_traffic.py joins the two passes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-traffic}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VOTENET_BENCH_NO_DP_LEG=1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/f -o p -- python3 $R/bench.py --headline-only --no-cpu-baseline --steps 12 --warmup 2 > $O/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/w -o p -- python3 $R/bench.py --headline-only --no-cpu-baseline --steps 12 --warmup 2 > $O/write.log 2>&1
python3 $R/tools/step_traffic.py $(find $O/f -name "*.db" | head -1) $(find $O/w -name "*.db" | head -1) 22 > $O/step_traffic.txt 2>&1
rm -rf $O/f $O/w
head -50 $O/step_traffic.txt | cut -c1-200
