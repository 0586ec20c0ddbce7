#!/bin/bash
# round 3, call h: strip kernel with dword boundary stores (config 4), saturation / frame-limit cases on the GPU
mkdir -p gpurun_out
line() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['phases_ms_per_step'], o['roofline']['kernel'][:24], o.get('parity',{}).get('mismatching_alignments'))" $1 $2; }
timeout 900 python -m pytest tests/test_saturation.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/h_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/h_pytest.log; tail -3 gpurun_out/h_pytest.log
timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/h_c4.log 2>&1; line gpurun_out/h_c4.log c4
