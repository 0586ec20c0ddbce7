# round 2, final evidence on the final kernel source: GPU suite, bench lines of configs 2-5, the PMC / kernel-trace passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 3 gpurun_out/pytest_gpu.log
bash scripts/gpu_profile_round2.sh 2>&1 | grep -E "rc=|hbm_bytes_per_alignment|config[245]\"" | head -40
mkdir -p profiles; cp gpurun_out/profiles_round2/round2_traffic.json profiles/round2_traffic.json      # (on the box: bench.py below reads it)
timeout 200 python bench.py > gpurun_out/final2_config2.log 2>&1; echo "config2 rc=$?"
timeout 200 python bench.py --config 3 > gpurun_out/final2_config3.log 2>&1; echo "config3 rc=$?"
timeout 200 python bench.py --config 4 > gpurun_out/final2_config4.log 2>&1; echo "config4 rc=$?"
timeout 300 python bench.py --config 5 > gpurun_out/final2_config5.log 2>&1; echo "config5 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final2_*.log")):
    try:
        d = json.loads([l for l in open(f).read().strip().split("\n") if l.startswith("{")][-1])
        print(f.split("final2_")[1][:-4], d["value"], d["ms_per_step"], d.get("phases_ms_per_step"), {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")},
              (d.get("roofline_valu") or {}).get("frac"), d["roofline"]["frac"], d["roofline"]["traffic"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "failed", e, open(f).read()[-400:])
PY
