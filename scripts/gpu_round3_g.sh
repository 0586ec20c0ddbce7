#!/bin/bash
# round 3, call g: duration of k_fill<10,frame> with and without the target staging work (rocprofv3 kernel trace; the measurement-only
# variant computes wrong scores, the batch call then fails in its locate pass -- after the fill launches this is about)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof3g; rm -rf $P; mkdir -p $P
cd /tmp
for v in base stagefree; do
  if [ $v = base ]; then L=$R/complete-striped-smith-waterman-library_amd/libssw.so; else L=$R/complete-striped-smith-waterman-library_amd/variants/libssw_$v.so; fi
  SSW_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats -d $P/$v -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --also none > $P/$v.log 2>&1
  python3 - $P/$v $v <<'PY'
import glob, sqlite3, sys
d = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)[0]
for r in sqlite3.connect(d).execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%k_fill%' group by name"):
    print(sys.argv[2], r)
PY
done > $R/gpurun_out/g_stagefree.txt 2>&1
cat $R/gpurun_out/g_stagefree.txt
rm -rf $P
