R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fps_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t -o s -- python3 $R/tools/pmc_fps.py > $O/run.log 2>&1
db=$(find $O/t -name "*.db" | head -1)
python3 $R/tools/rocpd_stats.py $db 20 > $O/stats.txt 2>&1
rm -rf $O/t
cut -c1-60,100-160 $O/stats.txt | head -16
