#!/bin/bash
# round 3, call r: config 4's strip fill with two (shipped build) and three wavefronts per SIMD (measurement-only build of
# scripts/probes/chainq_occupancy_variant.py: four-entry strip profile, 168 registers -- wrong scores, only the fill phase counts)
mkdir -p gpurun_out
V=$PWD/complete-striped-smith-waterman-library_amd
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'))" $1 $2 || tail -3 $1; }
for v in base chainq3; do
  if [ $v = base ]; then L=$V/libssw.so; else L=$V/variants/libssw_$v.so; fi
  SSW_LIB=$L timeout 150 python bench.py --config 4 --flag 0 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r_c4_$v.log 2>&1; short gpurun_out/r_c4_$v.log c4_flag0_$v
done
