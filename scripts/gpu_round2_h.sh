# round 2, call H: config 5 at full size with the streaming buffers kept in the context, the reference's own CLI on this library
# (wall time, 100 reads x 1 Mb: the shape of the reference's demo), the lane-model kernel's line, and one SQ counter pass of
# config 4 at full size (strip tickets; the reduced batch of the profile script runs with job tickets)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py --config 5 > gpurun_out/h_config5.log 2>&1; echo "config5 rc=$?"; grep "^{" gpurun_out/h_config5.log | tail -1 | cut -c1-200
timeout 200 python bench.py --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 > gpurun_out/h_literal.log 2>&1; echo "literal rc=$?"
python - <<'PY'
import numpy as np
z = np.load("tests/golden/chr3_1M.npz")
L = "ACGTN"
open("/tmp/1M.fa", "w").write(">chr3_1M\n" + "".join(L[c] for c in z["target"]) + "\n")
with open("/tmp/reads.fq", "w") as f:
    for i, r in enumerate(z["reads"]):
        s = "".join(L[c] for c in r); f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
PY
: > gpurun_out/h_dropin_time.log
for exe in oracle/_ref/ssw_test_dropin complete-striped-smith-waterman-library_amd/ssw_test_gpu; do
  for rep in 1 2; do
    s=$(date +%s.%N); timeout 120 $exe -c /tmp/1M.fa /tmp/reads.fq > /tmp/out_$rep.txt 2> /tmp/err_$rep.txt; rc=$?; e=$(date +%s.%N)
    echo "$exe -c 1M.fa reads.fq (100 reads x 1 Mb target, run $rep): rc $rc, $(python -c "print('%.3f' % ($e - $s))") s wall, $(grep -c optimal_alignment_score /tmp/out_$rep.txt) alignments printed" >> gpurun_out/h_dropin_time.log
  done
done
cmp /tmp/out_1.txt /tmp/out_2.txt && echo "stdout of the two front-ends' last runs identical: $(md5sum < /tmp/out_2.txt)" >> gpurun_out/h_dropin_time.log
cat gpurun_out/h_dropin_time.log
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/prof2/pmc4full_sq1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof2/pmc4full_sq1.log 2>&1; echo "pmc4full rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
for d in glob.glob("gpurun_out/prof2/pmc4full_sq1/**/*results.db", recursive=True):
    c = sqlite3.connect(d)
    with open("gpurun_out/h_config4_fullsize_pmc.csv", "w") as f:
        f.write("# rocprofv3 --pmc SQ_* -- python bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0  (full size: 5000 pairs, strip tickets; two passes over the batch)\n")
        f.write("kernel,counter,dispatches,sum,avg_per_dispatch,avg_dispatch_ns\n")
        for r in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection where kernel_name like '%k_%' group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write("\"%s\",%s,%d,%.6g,%.6g,%.0f\n" % r)
    print(open("gpurun_out/h_config4_fullsize_pmc.csv").read()[:3000])
PY
find gpurun_out/prof2 -name "*.db" -size +30M -delete
