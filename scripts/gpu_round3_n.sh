#!/bin/bash
# round 3, call n: the bench lines of the evidence run again (exclusive scratch budget when every rank has its own GPU); tests + rocprofv3 passes
# of gpu_round3_final.sh are unchanged (same kernel source)
mkdir -p gpurun_out
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['n_gpus'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'), (o.get('cpu_baseline') or {}).get('value'))" $1 $2; }
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/final_c2.log 2>&1; short gpurun_out/final_c2.log c2
timeout 300 python bench.py --flag 2 --steps 3 --warmup 1 --cpu-sample 0 --also none > gpurun_out/final_c2_flag2.log 2>&1; short gpurun_out/final_c2_flag2.log c2_flag2
timeout 300 python bench.py --config 3 --steps 3 --warmup 1 > gpurun_out/final_c3.log 2>&1; short gpurun_out/final_c3.log c3
timeout 300 python bench.py --config 4 --steps 3 --warmup 1 > gpurun_out/final_c4.log 2>&1; short gpurun_out/final_c4.log c4
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/final_c5.log 2>&1; short gpurun_out/final_c5.log c5
timeout 600 python bench.py --gpus 2 --config 3 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/final_c3_2ranks_one_gpu.log 2>&1; short gpurun_out/final_c3_2ranks_one_gpu.log c3_2ranks
timeout 300 python bench.py --pool 2 --steps 2 --warmup 1 --cpu-sample 0 --also none > gpurun_out/final_c2_pool2.log 2>&1; short gpurun_out/final_c2_pool2.log c2_pool2
timeout 600 python -m pytest tests/test_threads_pool.py tests/test_abi.py -x -q -m gpu > gpurun_out/n_pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/n_pytest.log
