#!/bin/bash
# round 3, call c: rows per lane of the strip kernel under the frame form (config 4), k_literal with its state in LDS
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gap_regime" > gpurun_out/c_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c_pytest.log; tail -3 gpurun_out/c_pytest.log
timeout 200 python bench.py --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 --also none > gpurun_out/c_literal.log 2>&1; tail -1 gpurun_out/c_literal.log | cut -c1-200
for xr in 8 10 12; do
  SSW_GPU_XR=$xr timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/c_c4_xr$xr.log 2>&1; tail -1 gpurun_out/c_c4_xr$xr.log | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('xr', $xr, o['value'], o['phases_ms_per_step'], o['roofline']['kernel'], o['parity']['mismatching_alignments'])"
done
