#!/bin/bash
# round 3, call i: k_literal with chunked prefetch (8 segments per round trip) and the matrix in LDS
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_threads_pool.py -x -q -m gpu -k "gap_regime or pool or threads or busy" > gpurun_out/i_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/i_pytest.log; tail -3 gpurun_out/i_pytest.log
timeout 200 python bench.py --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 --also none > gpurun_out/i_literal.log 2>&1
python3 -c "import sys,json; o=json.loads([l for l in open('gpurun_out/i_literal.log') if l.startswith('{')][-1]); print('literal', o['value'], o['phases_ms_per_step'], o['parity']['mismatching_alignments'], o['cpu_baseline']['value'])"
timeout 200 python bench.py --reads 20000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 0 --cpu-sample 0 --also none > gpurun_out/i_literal20k.log 2>&1
python3 -c "import sys,json; o=json.loads([l for l in open('gpurun_out/i_literal20k.log') if l.startswith('{')][-1]); print('literal20k', o['value'], o['phases_ms_per_step'])"
