#!/bin/bash
# round 3, call o: k_fill / k_filldb with the out-ring stores of every lane (no branch per step) against the committed build, and the
# measurement-only build without per-column maxima (scripts/probes/no_streams_variant.py: wrong results, only k_fill's duration counts)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof3o; rm -rf $P; mkdir -p $P
V=$R/complete-striped-smith-waterman-library_amd
cd /tmp
for v in base uncond nostreams; do
  if [ $v = base ]; then L=$V/libssw.so; else L=$V/variants/libssw_$v.so; fi
  SSW_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats -d $P/$v -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --also none > $P/$v.log 2>&1
  python3 - $P/$v $v <<'PY'
import glob, sqlite3, sys
d = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
for r in sqlite3.connect(d).execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%k_fill%' group by name"):
    print(sys.argv[2], r)
PY
done > $R/gpurun_out/o_kfill.txt 2>&1
cat $R/gpurun_out/o_kfill.txt
cd $R
short() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o.get('phases_ms_per_step'), (o.get('parity') or {}).get('mismatching_alignments'))" $1 $2; }
for v in base uncond; do
  if [ $v = base ]; then L=$V/libssw.so; else L=$V/variants/libssw_$v.so; fi
  SSW_LIB=$L timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --also none > gpurun_out/o_c2_$v.log 2>&1; short gpurun_out/o_c2_$v.log c2_$v
  SSW_LIB=$L timeout 200 python bench.py --config 5 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/o_c5_$v.log 2>&1; short gpurun_out/o_c5_$v.log c5_$v
  SSW_LIB=$L timeout 200 python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/o_c3_$v.log 2>&1; short gpurun_out/o_c3_$v.log c3_$v
done
rm -rf $P
