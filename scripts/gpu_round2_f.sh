# round 2, call F: where does k_filldb's time go (kernel trace per size class, with and without the chain-best filter), and the
# traceback rounds of config 4 with timestamps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_f
C5="--config 5 --reads 8192 --db-targets 2048 --db-chunk 2048 --steps 1 --warmup 1 --cpu-sample 0"
timeout 120 python bench.py $C5 > gpurun_out/f_config5_on.log 2>&1; tail -c 400 gpurun_out/f_config5_on.log | cut -c1-400
SSW_GPU_DB_CHAIN_BEST=0 timeout 120 python bench.py $C5 > gpurun_out/f_config5_off.log 2>&1; tail -c 400 gpurun_out/f_config5_off.log | cut -c1-400
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_f/trace5 -o bench -- python $GRAFT_REPO_ROOT/bench.py $C5 > $GRAFT_REPO_ROOT/gpurun_out/prof_f/trace5.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
for d in glob.glob("gpurun_out/prof_f/trace5/**/*results.db", recursive=True):
    db = sqlite3.connect(d)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), max(vgpr_count), max(lds_size), max(grid_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open("gpurun_out/f_config5_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,percent,vgpr,lds_bytes,max_grid_x\n")
        for r in rows:
            f.write("\"%s\",%d,%d,%.0f,%.3f,%d,%d,%d\n" % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5], r[6]))
    print(open("gpurun_out/f_config5_kernel_stats.csv").read()[:2600])
PY
SSW_GPU_DEBUG=1 SSW_GPU_XR=8 timeout 150 python bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/f_config4_debug.log 2>&1; grep -v "chainq\|resident" gpurun_out/f_config4_debug.log | cut -c1-260 | head -40
