#!/bin/bash
# round 3, call d: compile-time variants of k_filldb (step-loop unrolling, register budget) and of the strip kernel's unrolling; k_literal
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd/variants
line() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['phases_ms_per_step'], o['roofline']['kernel'][:24], o.get('parity',{}).get('mismatching_alignments'))" $1 $2; }
timeout 200 python bench.py --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 --also none > gpurun_out/d_literal.log 2>&1; line gpurun_out/d_literal.log literal
timeout 200 python bench.py --config 5 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/d_c5_base.log 2>&1; line gpurun_out/d_c5_base.log c5_base
for v in dbu1 dbu4 dbw4; do
  SSW_LIB=$V/libssw_$v.so timeout 200 python bench.py --config 5 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/d_c5_$v.log 2>&1; line gpurun_out/d_c5_$v.log c5_$v
done
SSW_LIB=$V/libssw_su1.so timeout 200 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/d_c4_su1.log 2>&1; line gpurun_out/d_c4_su1.log c4_su1
