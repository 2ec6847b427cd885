#!/bin/bash
# three concurrent pk_opsel_hazard processes (the GPU shared between them).  usage: pk3.sh [launches] [side]
b=tools/probe/bin/pk_opsel_hazard
$b ${1:-100} ${2:-0} > /tmp/p1.txt 2>&1 &
p1=$!
$b ${1:-100} ${2:-0} > /tmp/p2.txt 2>&1 &
p2=$!
$b ${1:-100} ${2:-0} > /tmp/p3.txt 2>&1
wait $p1 $p2
cat /tmp/p1.txt /tmp/p2.txt /tmp/p3.txt
