# round 2, call K: the database-search kernel with the f16 form first / the int16 repeat, explicit LDS pipelining and the deferred
# record branch -- compile-time variants (scripts/build_variants.sh) x chains per workgroup on the config-5 shape, and the
# deferred branch in the strip kernel on config 4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd/variants
timeout 200 python -m pytest tests/test_search_db.py tests/test_saturation.py -m gpu -x -q > gpurun_out/k_pytest_db.log 2>&1; echo "pytest db rc=$?"; tail -n 3 gpurun_out/k_pytest_db.log
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments", "queries_with_wrong_checksum")}, (d.get("roofline_valu") or {}).get("frac"), (d.get("roofline") or {}).get("kernel"))
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-500:])
PY
}
c5() { # name, lib ("" = default), env...
	name=$1; lib=$2; shift 2
	( [ -n "$lib" ] && export SSW_LIB=$PWD/$V/libssw_$lib.so; for e in "$@"; do export "$e"; done
	  timeout 100 python bench.py --config 5 --reads 8192 --db-targets 2048 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/k_c5_$name.log 2>&1 )
	show gpurun_out/k_c5_$name.log "c5shape $name"
}
c5 v1_n32 "" SSW_GPU_DB_CHAINS=32
c5 v1_n16 "" SSW_GPU_DB_CHAINS=16
c5 v1_n32_int16 "" SSW_GPU_DB_CHAINS=32 SSW_GPU_DB_F16=0
c5 v1_n16_int16 "" SSW_GPU_DB_CHAINS=16 SSW_GPU_DB_F16=0
c5 v0_n32 v0 SSW_GPU_DB_CHAINS=32
c5 v0_n16 v0 SSW_GPU_DB_CHAINS=16
c5 v2_n32 v2 SSW_GPU_DB_CHAINS=32
c5 v2_n16 v2 SSW_GPU_DB_CHAINS=16
c5 v5_n32 v5 SSW_GPU_DB_CHAINS=32
c5 v3_nodefer_n32 v3 SSW_GPU_DB_CHAINS=32
c5 v3_nodefer_n16 v3 SSW_GPU_DB_CHAINS=16
c5 v4_notrack_n32 v4 SSW_GPU_DB_CHAINS=32
c4() { name=$1; lib=$2; shift 2
	( [ -n "$lib" ] && export SSW_LIB=$PWD/$V/libssw_$lib.so; for e in "$@"; do export "$e"; done
	  timeout 150 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/k_c4_$name.log 2>&1 )
	show gpurun_out/k_c4_$name.log "config4 $name"
}
c4 deferred ""
c4 nodefer v3
c4 notrack v4
