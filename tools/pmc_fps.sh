#!/bin/bash
# GPU box: the two counter passes behind roofline.traffic (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 --pmc runs of tools/pmc_fps.py,
# no tracing domains beside them).  Tables land in gpurun_out/$1; `python tools/make_profile_summaries.py $1 rNN` composes
# profiles/rNN_pmc_fps.txt and refreshes profiles/pmc_latest.json afterwards.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-pmcfps}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/fps_fetch -o p -- python3 $R/tools/pmc_fps.py > $O/fps_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/fps_write -o p -- python3 $R/tools/pmc_fps.py > $O/fps_write.log 2>&1
cd $R
for d in fps_fetch fps_write; do python tools/rocpd_pmc.py $(find $O/$d -name "*.db" | head -1) > $O/${d}.txt 2>&1; done
rm -rf $O/fps_fetch $O/fps_write
cat $O/fps_fetch.txt | head -12
