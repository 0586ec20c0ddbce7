#!/bin/bash
# round 3, call a: the column-frame form of k_fill on the MI355X -- selftest, parity tests of the short-query path, config 2/3 rates in
# the frame form and (SSW_GPU_FILL_FORM=1) the f16 form of round 2 for comparison
mkdir -p gpurun_out
export PYTHONPATH=tests:complete-striped-smith-waterman-library_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_saturation.py tests/test_full_size.py -x -q -m gpu -k "not config4 and not config5 and not long_read and not database" > gpurun_out/a_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/a_pytest.log
tail -5 gpurun_out/a_pytest.log
timeout 300 python bench.py --steps 3 --warmup 1 --also none > gpurun_out/a_c2_frame.log 2>&1; tail -1 gpurun_out/a_c2_frame.log | cut -c1-400
SSW_GPU_FILL_FORM=1 timeout 300 python bench.py --steps 3 --warmup 1 --also none --cpu-sample 0 > gpurun_out/a_c2_f16.log 2>&1; tail -1 gpurun_out/a_c2_f16.log | cut -c1-400
timeout 300 python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/a_c3_frame.log 2>&1; tail -1 gpurun_out/a_c3_frame.log | cut -c1-400
