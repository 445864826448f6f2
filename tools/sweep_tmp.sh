set -e
S=bunny,bunny_obj,balls2,dragon_large_tris,iceBot
for i in 1 2 3; do
RT_LIBRARY=raytracer_amd/libbase.so timeout -k 10 120 python tools/gpu_one.py $S >> gpurun_out/sweep.log 2>&1
timeout -k 10 120 python tools/gpu_one.py $S >> gpurun_out/sweep.log 2>&1
done
