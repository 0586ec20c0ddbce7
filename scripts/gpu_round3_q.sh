#!/bin/bash
# round 3, call q: run_strip (k_chainq) with F's hand-off fused with the last row's "- gapE" and three-word boundary records where the
# 8-bit rule's row mask is not in play, against the build before (variants/libssw_base.so); the whole GPU suite on the new kernels
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/q_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/q_pytest.log; tail -3 gpurun_out/q_pytest.log
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'))" $1 $2; }
for v in base new; do
  if [ $v = base ]; then L=$PWD/$V/variants/libssw_base.so; else L=$PWD/$V/libssw.so; fi
  SSW_LIB=$L timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/q_c4_$v.log 2>&1; short gpurun_out/q_c4_$v.log c4_$v
done
