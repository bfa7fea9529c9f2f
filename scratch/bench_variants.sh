#!/bin/bash
# every option of bench.py once, on small inputs: none may fail, every line must parse
OUT=gpurun_out/benchv; rm -rf $OUT; mkdir -p $OUT
i=0
while read -r args; do
  i=$((i+1))
  if timeout -k 10 200 python3 bench.py $args > $OUT/$i.json 2> $OUT/$i.err; then
    python3 -c "import json,sys; d=json.load(open('$OUT/$i.json')); print('ok  ', '$args', '->', round(d['value']), d['roofline']['kernel'], round(d['roofline']['frac'],3))"
  else
    echo "FAIL $args"; tail -3 $OUT/$i.err
  fi
done <<ARGS
--steps 9 --warmup 1 --lattice 300,300,1 --cpu-seconds 0
--steps 9 --warmup 1 --lattice 500,400,1 --cpu-seconds 0 --vector-kind z4
--steps 7 --warmup 2 --lattice 60,50,40 --cpu-seconds 0 --model dwave
--steps 7 --warmup 2 --lattice 90,90,90 --cpu-seconds 0 --model dwave --vectors-per-gpu 4
--steps 16 --warmup 2 --lattice 700,700,1 --cpu-seconds 0 --vectors-per-gpu 3
--steps 16 --warmup 2 --lattice 200,200,1 --cpu-seconds 2 --vectors-per-gpu 64
--steps 8 --warmup 1 --lattice 64,64,64 --cpu-seconds 0 --mode slab
--steps 8 --warmup 1 --lattice 400,400,1 --cpu-seconds 0 --lanes 8
--steps 8 --warmup 1 --lattice 1000,1000,1 --cpu-seconds 0 --temperature 0.1
ARGS
