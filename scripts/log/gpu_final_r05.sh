mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r05f/gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05f/smoke.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05f/bench_driver_args.json 2> gpurun_out/r05f/bench.err
tail -3 gpurun_out/r05f/gpu_suite.txt; tail -1 gpurun_out/r05f/smoke.txt; cut -c1-300 gpurun_out/r05f/bench_driver_args.json
