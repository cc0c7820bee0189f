run() { HODOR_LIB=$PWD/$1 python bench.py --mode sixstep --steps 60 --warmup 20 --no-cpu-baseline --no-extra --soak-seconds 0 --allow-knobs 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3; do echo "base $(run hodor_amd/libhodor_gpu_base.so)  r04a $(run hodor_amd/libhodor_gpu_r04a.so)  now $(run hodor_amd/libhodor_gpu.so)"; done
