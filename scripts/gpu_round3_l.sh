#!/bin/bash
# round 3, call l: default scratch budget sized for 288 GB (min(200 GiB, 2/3 of free)) with halving on allocation failure: default line with
# `also`, two ranks / two pool workers sharing the GPU, config 4 (traceback scratch next to the fill's buffers), pool + budget tests
mkdir -p gpurun_out
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['n_gpus'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'), {k:(v.get('value'), (v.get('parity') or {}).get('mismatching_alignments')) for k,v in (o.get('also') or {}).items() if isinstance(v,dict)})" $1 $2; }
timeout 600 python -m pytest tests/test_threads_pool.py tests/test_full_size.py tests/test_search_db.py -x -q -m gpu > gpurun_out/l_pytest.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/l_pytest.log
timeout 600 python bench.py --steps 6 --warmup 1 --cpu-sample 0 > gpurun_out/l_c2.log 2>&1; short gpurun_out/l_c2.log c2
timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/l_c3.log 2>&1; short gpurun_out/l_c3.log c3
timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/l_c4.log 2>&1; short gpurun_out/l_c4.log c4
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/l_c2_2ranks.log 2>&1; short gpurun_out/l_c2_2ranks.log c2_2ranks; grep -c "allocation" gpurun_out/l_c2_2ranks.log
timeout 300 python bench.py --pool 2 --steps 2 --warmup 1 --cpu-sample 0 --also none > gpurun_out/l_c2_pool2.log 2>&1; short gpurun_out/l_c2_pool2.log c2_pool2
