#!/usr/bin/env bash
# probe (GPU box): where the time of the piece-layout (half.hip) kernels goes.  Per argument "<file> <flags>" (e.g. "mlp_fast -DBF3_ABL=2",
# "pool_bwd -DGRAM_ABL=1") the object is rebuilt with the flags, ONE level's forward + backward runs alone on one stream under
# rocprofv3 --kernel-trace (tools/pmc_pieces.py: real geometry of room scenes), and the average durations of the two kernels the verdict
# names -- the pooled forward GEMM mlp_linear_fast_kernel<2,2,2,2,0,8,true> (round 4: <...,0,2,true>) and gram_bf3_kernel -- are printed.  Ablated builds are
# wrong by construction: only the time is read.  Leaves the library in its DEFAULT build.
#   BF3_ABL / GRAM_ABL: 1 no MFMAs, 2 no epilogue, 4 no global loads after the prologue, 8 no staging after the prologue
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
C=$R/votenet_amd/csrc; O=$R/gpurun_out/ablate_half; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
LEVELS=${LEVELS:-"sa2 sa1"}
cd /tmp && export TMPDIR=/tmp
one() { # file flags tag
  /opt/rocm/bin/hipcc $FLAGS $2 -c $C/$1.hip -o $C/obj/$1.o 2>/dev/null || { echo "build failed: $1 $2"; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/obj/*.o -o $R/votenet_amd/lib/libvotenet_hip.so
  for lv in $LEVELS; do
    rm -rf $O/t; REP=6 timeout 300 rocprofv3 --kernel-trace -d $O/t -o p -- python3 $R/tools/pmc_pieces.py $lv > $O/run.log 2>&1
    db=$(find $O/t -name "*.db" | head -1)
    python3 $R/tools/rocpd_stats.py $db 60 2>/dev/null | grep -E "mlp_linear_fast_kernel<2, 2, 2, 2, 0, [28], true>|gram_bf3_kernel|mlp_linear_fast_kernel<4, 1, 1, 2, 5, 7, true>|mlp_linear_fast_kernel<4, 1, 1, 2, 3, 0, true>" | \
      awk -v t="$3" -v l=$lv '{n=$1; for(i=2;i<=NF-9;i++) n=n" "$i; printf "%-14s %-4s %-52s calls %3s avg_us %8s\n", t, l, n, $(NF-8), $(NF-6)}'
    rm -rf $O/t
  done
}
one mlp_fast "" base
one pool_bwd "" base
for spec in "$@"; do
  set -- $spec; f=$1; shift
  one $f "$*" "$*"
  one $f "" "(restored)" > /dev/null
done
