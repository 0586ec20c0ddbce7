#!/bin/bash
# round 3, call e: GPU tests of the batched bindings / pool / database-search changes; k_fill with the adds of a step hoisted into one run
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd/variants
line() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['phases_ms_per_step'], o['roofline']['kernel'][:24], o.get('parity',{}).get('mismatching_alignments'))" $1 $2; }
timeout 900 python -m pytest tests/test_wrappers.py tests/test_search_db.py tests/test_threads_pool.py tests/test_abi.py -x -q -m gpu > gpurun_out/e_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/e_pytest.log; tail -4 gpurun_out/e_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --also none > gpurun_out/e_c2_base.log 2>&1; line gpurun_out/e_c2_base.log c2_base
SSW_LIB=$V/libssw_hoist.so timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --also none > gpurun_out/e_c2_hoist.log 2>&1; line gpurun_out/e_c2_hoist.log c2_hoist
SSW_LIB=$V/libssw_hoist.so timeout 300 python bench.py --config 5 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/e_c5_hoist.log 2>&1; line gpurun_out/e_c5_hoist.log c5_hoist
