#!/bin/bash
# three concurrent tools/probe/h2_repeat.py processes on one GPU (plus each one's own MFMA side stream); prints each one's verdict
n=${1:-300}
SEED=1 python tools/probe/h2_repeat.py $n > /tmp/h1.txt 2>&1 &
p1=$!
SEED=2 python tools/probe/h2_repeat.py $n > /tmp/h2.txt 2>&1 &
p2=$!
SEED=3 python tools/probe/h2_repeat.py $n > /tmp/h3.txt 2>&1
wait $p1 $p2
for f in /tmp/h1.txt /tmp/h2.txt /tmp/h3.txt; do tail -n 1 $f; done
