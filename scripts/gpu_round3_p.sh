#!/bin/bash
# round 3, call p: the two lane hand-offs fused with the arithmetic behind them (v_max_u32_dpp / v_sub_u32_dpp, k_fill and k_filldb) against
# the build before (variants/libssw_base.so), and the whole GPU suite on the new kernels
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/p_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/p_pytest.log; tail -3 gpurun_out/p_pytest.log
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'))" $1 $2; }
for v in base new; do
  if [ $v = base ]; then L=$PWD/$V/variants/libssw_base.so; else L=$PWD/$V/libssw.so; fi
  SSW_LIB=$L timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --also none > gpurun_out/p_c2_$v.log 2>&1; short gpurun_out/p_c2_$v.log c2_$v
  SSW_LIB=$L timeout 200 python bench.py --config 5 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/p_c5_$v.log 2>&1; short gpurun_out/p_c5_$v.log c5_$v
  SSW_LIB=$L timeout 200 python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/p_c3_$v.log 2>&1; short gpurun_out/p_c3_$v.log c3_$v
done
