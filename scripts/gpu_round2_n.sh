# round 2, call N: cleaned-up k_filldb (f16 first, bfi records), score1 floor in the window passes of the strip kernel: GPU suite + configs 4, 5
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/n_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/n_pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")}, (d.get("roofline_valu") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:30])
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-500:])
PY
}
timeout 150 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/n_c4.log 2>&1; show gpurun_out/n_c4.log "config4"
SSW_GPU_DEBUG=1 timeout 150 python bench.py --config 4 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/n_c4_debug.log 2>&1; grep -i -E "retry|window|round" gpurun_out/n_c4_debug.log | head -12
timeout 100 python bench.py --config 5 --reads 8192 --db-targets 2048 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/n_c5.log 2>&1; show gpurun_out/n_c5.log "c5shape"
timeout 100 python bench.py --config 2 --flag 2 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/n_c2f2.log 2>&1; show gpurun_out/n_c2f2.log "config2 flag2"
