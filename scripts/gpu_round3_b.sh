#!/bin/bash
# round 3, call b: column-frame form in all three fill kernels -- the whole GPU suite, then configs 2 (with `also`), 4 and 5
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/b_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/b_pytest.log
tail -4 gpurun_out/b_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/b_c2.log 2>&1; tail -1 gpurun_out/b_c2.log | cut -c1-300
timeout 300 python bench.py --config 4 --steps 2 --warmup 1 > gpurun_out/b_c4.log 2>&1; tail -1 gpurun_out/b_c4.log | cut -c1-300
timeout 300 python bench.py --config 5 --steps 1 --warmup 1 > gpurun_out/b_c5.log 2>&1; tail -1 gpurun_out/b_c5.log | cut -c1-300
