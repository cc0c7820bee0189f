bash bench/profile.sh r02 > gpurun_out/profile_r02.log 2>&1
tail -5 gpurun_out/profile_r02.log
bash bench/ablate.sh 2>&1 | tee gpurun_out/ablate_r02.txt
python bench.py > gpurun_out/bench_default_r02.json 2> gpurun_out/bench_default_r02.err; tail -c 600 gpurun_out/bench_default_r02.json
python bench.py --mode sixstep --no-cpu-baseline --no-extra > gpurun_out/bench_sixstep_w1_r02.json 2>/dev/null
python bench/ali_replay.py 20 8 > gpurun_out/ali_replay_r02.txt 2>&1
python bench/fri_sizes.py > gpurun_out/fri_sizes_r02.txt 2>&1
python bench/pointwise.py > gpurun_out/pointwise_r02.txt 2>&1
python bench/slice_api.py > gpurun_out/slice_api_r02.txt 2>&1
python bench/size_sweep.py --out gpurun_out/size_sweep.json > gpurun_out/size_sweep_r02.log 2>&1; tail -1 gpurun_out/size_sweep_r02.log
