#!/bin/bash
# round 3, call k: rows per lane of the window passes under the frame form (config 4)
mkdir -p gpurun_out
line() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['phases_ms_per_step'], o.get('parity',{}).get('mismatching_alignments'))" $1 $2; }
for xw in 6 10 12; do
  SSW_GPU_XR_WINDOW=$xw timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/k_c4_xw$xw.log 2>&1; line gpurun_out/k_c4_xw$xw.log c4_xw$xw
done
