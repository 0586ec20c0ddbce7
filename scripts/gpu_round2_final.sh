# round 2, final evidence run (everything under its own time limit; outputs in gpurun_out/, summaries copied to profiles/):
#   GPU test-suite, bench lines of BASELINE configs 2-5 at their stated sizes, the in-library work queues (two workers on the one
#   device), a 2-rank torchrun of the config-3 mode on the one device, the lane-model kernel (gapO <= gapE), the reference's own CLI
#   on this library (wall time on the 1 Mb demo shape), then the rocprofv3 passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 4 gpurun_out/pytest_gpu.log
timeout 200 python bench.py > gpurun_out/final_config2.log 2>&1; echo "config2 rc=$?"
timeout 200 python bench.py --config 3 > gpurun_out/final_config3.log 2>&1; echo "config3 rc=$?"
timeout 200 python bench.py --config 4 > gpurun_out/final_config4.log 2>&1; echo "config4 rc=$?"
timeout 300 python bench.py --config 5 > gpurun_out/final_config5.log 2>&1; echo "config5 rc=$?"
timeout 200 python bench.py --config 2 --flag 2 --steps 1 --cpu-sample 0 > gpurun_out/final_config2_flag2.log 2>&1; echo "config2 flag2 rc=$?"
timeout 200 python bench.py --pool 2 --steps 1 --cpu-sample 0 > gpurun_out/final_pool2.log 2>&1; echo "pool rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --config 3 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/final_config3_2ranks_one_gpu.log 2>&1; echo "2 ranks rc=$?"
timeout 200 python bench.py --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 > gpurun_out/final_literal.log 2>&1; echo "literal rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final_*.log")):
    try:
        d = json.loads([l for l in open(f).read().strip().split("\n") if l.startswith("{")][-1])
        print(f.split("final_")[1][:-4], d["value"], d["ms_per_step"], d.get("phases_ms_per_step"), {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")},
              (d.get("roofline_valu") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e, open(f).read()[-400:])
PY
# the reference's own CLI (src/main.c, built unmodified against libssw.so) on the shape of its demo: 100 reads x a 1 Mb target
python - <<'PY'
import numpy as np
z = np.load("tests/golden/chr3_1M.npz")
L = "ACGTN"
open("/tmp/1M.fa", "w").write(">chr3_1M\n" + "".join(L[c] for c in z["target"]) + "\n")
with open("/tmp/reads.fq", "w") as f:
    for i, r in enumerate(z["reads"]):
        s = "".join(L[c] for c in r); f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
PY
for exe in oracle/_ref/ssw_test_dropin complete-striped-smith-waterman-library_amd/ssw_test_gpu; do
  for rep in 1 2; do /usr/bin/time -f "$exe -c 1M.fa reads.fq (run $rep): %e s wall" timeout 120 $exe -c /tmp/1M.fa /tmp/reads.fq > /tmp/out_$rep.txt 2>> gpurun_out/final_dropin_time.log; done
done
grep -v "CPU time" gpurun_out/final_dropin_time.log | tail -n 6
bash scripts/gpu_profile_round2.sh 2>&1 | tail -n 45
