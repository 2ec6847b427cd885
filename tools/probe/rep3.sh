#!/bin/bash
# three concurrent scatter_repeat.py processes (the GPU shared between them); prints each one's last line.  usage: rep3.sh [steps]
n=${1:-20}
python tools/probe/scatter_repeat.py $n > /tmp/r1.txt 2>&1 &
p1=$!
python tools/probe/scatter_repeat.py $n > /tmp/r2.txt 2>&1 &
p2=$!
python tools/probe/scatter_repeat.py $n > /tmp/r3.txt 2>&1
wait $p1 $p2
for f in /tmp/r1.txt /tmp/r2.txt /tmp/r3.txt; do tail -n 1 $f; done
