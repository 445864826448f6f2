#!/bin/bash
# A/B timing of two builds of the library: ab.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
cp raytracer_amd/libraytracer_amd.so /tmp/orig.so
for i in $(seq 1 $R); do
  for L in $A $B; do
    cp raytracer_amd/$L raytracer_amd/libraytracer_amd.so
    timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-fast-bvh > gpurun_out/ab.log 2>&1 || exit 1
    python -c "
import json,sys;d=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1]);print(sys.argv[1],d['ms_per_step'],d['roofline'].get('avg_launch_ms'))" $L
  done
done
cp /tmp/orig.so raytracer_amd/libraytracer_amd.so
