# Developer probe (GPU box): the several-workers path of bench.py on one GPU.
set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -x -q -k "multi or workers or in_place or copies or two_ranks or bench" --timeout 300 --timeout-method thread > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
RT_BENCH_DEVICES=0,0 timeout -k 10 120 python bench.py --gpus 2 --steps 50 --warmup 25 --check > gpurun_out/r3_m2.json 2> gpurun_out/r3_m2.err
RT_MULTI_TEST_PEERS=1 RT_BENCH_DEVICES=0,0 timeout -k 10 120 python bench.py --gpus 2 --steps 50 --warmup 25 --check > gpurun_out/r3_m2p.json 2> gpurun_out/r3_m2p.err
RT_BENCH_DEVICES=0,0,0,0 timeout -k 10 120 python bench.py --gpus 4 --steps 50 --warmup 25 --check > gpurun_out/r3_m4.json 2> gpurun_out/r3_m4.err
RT_BENCH_DEVICES=0 timeout -k 10 120 python bench.py --gpus 1 --steps 50 --warmup 25 --check > gpurun_out/r3_m1.json 2> gpurun_out/r3_m1.err
