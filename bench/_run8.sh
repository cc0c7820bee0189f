python -m pytest tests/test_gpu_ali_replay.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "fri or config3 or composed or prover" 2>&1 | tail -3
for r in 1 2; do for f in 2 1; do HODOR_FRI_FUSE_FOLD=$f python bench/fri_sizes.py 2>&1 | tail -4 | tr '\n' ' '; echo " FUSE=$f"; done; done
python bench/ali_replay.py 20 8
python bench/size_sweep.py --out gpurun_out/size_sweep.json 2>&1 | tail -3
