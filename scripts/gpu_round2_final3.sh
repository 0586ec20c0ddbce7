# round 2, final evidence on the final kernel source (after the k_filldb / strip-kernel work of calls K-O): GPU suite, the PMC /
# kernel-trace passes, bench lines of configs 2-5 at their stated sizes + the side lines (flag 2, in-library work queues, two
# ranks on the one device, the lane-model kernel), SQ counters of config 4 at full size
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 3 gpurun_out/pytest_gpu.log
bash scripts/gpu_profile_round2.sh 2>&1 | grep -E "rc=|hbm_bytes_per_alignment|config[245]\"" | head -40
mkdir -p profiles; cp gpurun_out/profiles_round2/round2_traffic.json profiles/round2_traffic.json      # (on the box: bench.py below reads it)
timeout 200 python bench.py > gpurun_out/final3_config2.log 2>&1; echo "config2 rc=$?"
timeout 200 python bench.py --config 3 > gpurun_out/final3_config3.log 2>&1; echo "config3 rc=$?"
timeout 200 python bench.py --config 4 > gpurun_out/final3_config4.log 2>&1; echo "config4 rc=$?"
timeout 300 python bench.py --config 5 > gpurun_out/final3_config5.log 2>&1; echo "config5 rc=$?"
timeout 200 python bench.py --config 2 --flag 2 --steps 1 --cpu-sample 0 > gpurun_out/final3_config2_flag2.log 2>&1; echo "config2 flag2 rc=$?"
timeout 200 python bench.py --pool 2 --steps 1 --cpu-sample 0 > gpurun_out/final3_pool2.log 2>&1; echo "pool rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --config 3 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/final3_config3_2ranks_one_gpu.log 2>&1; echo "2 ranks rc=$?"
timeout 200 python bench.py --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 > gpurun_out/final3_literal.log 2>&1; echo "literal rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final3_*.log")):
    try:
        d = json.loads([l for l in open(f).read().strip().split("\n") if l.startswith("{")][-1])
        print(f.split("final3_")[1][:-4], d["value"], d["ms_per_step"], d.get("phases_ms_per_step"), {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")},
              (d.get("roofline_valu") or {}).get("frac"), d["roofline"]["frac"], d["roofline"]["traffic"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e, open(f).read()[-400:])
PY
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/gpurun_out/prof2/pmc4full_sq1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/prof2/pmc4full_sq1.log 2>&1; echo "pmc4full rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
for d in glob.glob("gpurun_out/prof2/pmc4full_sq1/**/*results.db", recursive=True):
    c = sqlite3.connect(d)
    with open("gpurun_out/final3_config4_fullsize_pmc.csv", "w") as f:
        f.write("# rocprofv3 --pmc SQ_* -- python bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0  (full size: 5000 pairs, strip tickets; two passes over the batch)\n")
        f.write("kernel,counter,dispatches,sum,avg_per_dispatch,avg_dispatch_ns\n")
        for r in c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection where kernel_name like '%k_%' group by kernel_name, counter_name order by kernel_name, counter_name"):
            f.write("\"%s\",%s,%d,%.6g,%.6g,%.0f\n" % r)
    print(open("gpurun_out/final3_config4_fullsize_pmc.csv").read()[:1800])
PY
find gpurun_out/prof2 -name "*.db" -size +30M -delete
