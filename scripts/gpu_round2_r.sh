# round 2, call R: lanes 15 / TAP park the column maxima they finish (k_fill, k_filldb: no ror moves back to lane 0), k_fill held to
# 72 registers up to 10 rows per lane: GPU suite, configs 2, 5 (shape), 2 with flag 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/r_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 gpurun_out/r_pytest.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")}, (d.get("roofline_valu") or {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-500:])
PY
}
timeout 150 python bench.py --config 2 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r_c2.log 2>&1; show gpurun_out/r_c2.log config2
timeout 100 python bench.py --config 5 --reads 8192 --db-targets 2048 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r_c5.log 2>&1; show gpurun_out/r_c5.log c5shape
timeout 150 python bench.py --config 3 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r_c3.log 2>&1; show gpurun_out/r_c3.log config3
