# round 2, call L: k_filldb with best-cell records stored to scratch (no register snapshot): compile-time variants on the config-5 shape
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd/variants
timeout 200 python -m pytest tests/test_search_db.py tests/test_saturation.py -m gpu -x -q > gpurun_out/l_pytest_db.log 2>&1; echo "pytest db rc=$?"; tail -n 3 gpurun_out/l_pytest_db.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")}, (d.get("roofline_valu") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:24])
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-500:])
PY
}
c5() { name=$1; lib=$2; shift 2
	( [ -n "$lib" ] && export SSW_LIB=$PWD/$V/libssw_$lib.so; for e in "$@"; do export "$e"; done
	  timeout 100 python bench.py --config 5 --reads 8192 --db-targets 2048 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/l_c5_$name.log 2>&1 )
	show gpurun_out/l_c5_$name.log "c5shape $name"
}
c5 default ""
c5 default_n32 "" SSW_GPU_DB_CHAINS=32
c5 default_int16 "" SSW_GPU_DB_F16=0
c5 nocap w0
c5 unroll4 w1
c5 deferred w2
c5 notrack w3
c5 nochainbest "" SSW_GPU_DB_CHAIN_BEST=0
timeout 300 python bench.py --config 5 --cpu-sample 0 > gpurun_out/l_c5_full.log 2>&1; show gpurun_out/l_c5_full.log "config5 full"
