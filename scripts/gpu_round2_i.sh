# round 2, call I: config 3 after sizing the tiles from the pairs a launch covers; one SQ counter pass of config 4 at full size
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof2
timeout 200 python bench.py --config 3 > gpurun_out/i_config3.log 2>&1; echo "config3 rc=$?"; grep "^{" gpurun_out/i_config3.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms_per_step'], d['roofline_valu']['frac'], d['parity']['mismatching_alignments'], d['roofline']['traffic'])"
timeout 200 python bench.py > gpurun_out/i_config2.log 2>&1; echo "config2 rc=$?"; grep "^{" gpurun_out/i_config2.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms_per_step'], d['roofline_valu']['frac'], d['parity']['mismatching_alignments'], d['roofline']['traffic'])"
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/prof2/pmc4full_sq1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof2/pmc4full_sq1.log 2>&1; echo "pmc4full rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
for d in glob.glob("gpurun_out/prof2/pmc4full_sq1/**/*results.db", recursive=True):
    c = sqlite3.connect(d)
    with open("gpurun_out/i_config4_fullsize_pmc.csv", "w") as f:
        f.write("# rocprofv3 --pmc SQ_* -- python bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0  (full size: 5000 pairs, strip tickets; two passes over the batch)\n")
        f.write("kernel,counter,dispatches,sum,avg_per_dispatch,avg_dispatch_ns\n")
        for r in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection where kernel_name like '%k_%' group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write("\"%s\",%s,%d,%.6g,%.6g,%.0f\n" % r)
    print(open("gpurun_out/i_config4_fullsize_pmc.csv").read()[:2500])
PY
find gpurun_out/prof2 -name "*.db" -size +30M -delete
