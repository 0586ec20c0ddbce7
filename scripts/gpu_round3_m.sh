#!/bin/bash
# round 3, call m: shared-device cases under the 288-GB-sized default budget (two ranks on one GPU, configs 2 and 3; pool of 2), single-rank lines
mkdir -p gpurun_out
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['n_gpus'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'))" $1 $2 || tail -3 $1; }
timeout 600 python bench.py --gpus 2 --config 3 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/m_c3_2ranks.log 2>&1; short gpurun_out/m_c3_2ranks.log c3_2ranks
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/m_c2_2ranks.log 2>&1; short gpurun_out/m_c2_2ranks.log c2_2ranks
timeout 600 python bench.py --gpus 3 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/m_c2_3ranks.log 2>&1; short gpurun_out/m_c2_3ranks.log c2_3ranks
timeout 300 python bench.py --pool 2 --steps 2 --warmup 1 --cpu-sample 0 --also none > gpurun_out/m_c2_pool2.log 2>&1; short gpurun_out/m_c2_pool2.log c2_pool2
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --also none > gpurun_out/m_c2.log 2>&1; short gpurun_out/m_c2.log c2
timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/m_c3.log 2>&1; short gpurun_out/m_c3.log c3
