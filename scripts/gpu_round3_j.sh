#!/bin/bash
# round 3, call j: window passes (locate / reverse) of the 64-lane strip kernel in the column-frame form: long-read parity tests + config 4
mkdir -p gpurun_out
line() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['phases_ms_per_step'], o['roofline']['kernel'][:24], o.get('parity',{}).get('mismatching_alignments'))" $1 $2; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_saturation.py tests/test_full_size.py -x -q -m gpu -k "long or strip or config4 or sat_strip or traceback or window" > gpurun_out/j_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/j_pytest.log; tail -3 gpurun_out/j_pytest.log
timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/j_c4.log 2>&1; line gpurun_out/j_c4.log c4_frame_windows
SSW_GPU_WINDOW_INT16=1 timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/j_c4_int16.log 2>&1; line gpurun_out/j_c4_int16.log c4_int16_windows
