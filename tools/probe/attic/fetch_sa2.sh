R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_pieces; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/p -o p -- python3 $R/tools/pmc_pieces.py sa2 > $O/x.log 2>&1
db=$(find $O/p -name "*.db" | head -1); python3 $R/tools/rocpd_pmc.py $db 2>&1 | grep "5, 6\|4, 0, true\|wgrad_fast_kernel<3\|sorted" | cut -c1-120; rm -rf $O/p
