#!/bin/bash
# usage: sweep.sh OUTFILE "env assignments" bench-args...
out=$1; shift; envs=$1; shift
env $envs timeout -k 10 300 python3 bench.py --cpu-seconds 0 "$@" >> $out 2>&1 || echo "bench failed: $envs $@" >> $out
