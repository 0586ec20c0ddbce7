# round 2, call O: rows per lane of the strip kernel again (the deferred record branch changed the per-step overhead)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1])
    print(sys.argv[2], d["value"], d["phases_ms_per_step"], {k: v for k, v in (d.get("parity") or {}).items() if k in ("sample", "mismatching_alignments")}, (d.get("roofline_valu") or {}).get("frac"), (d.get("roofline") or {}).get("kernel", "")[:40])
except Exception as e:
    print(sys.argv[2], "failed", e); print(open(sys.argv[1]).read()[-500:])
PY
}
for xr in 12 9 11 8; do
	SSW_GPU_XR=$xr timeout 150 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/o_c4_xr$xr.log 2>&1; show gpurun_out/o_c4_xr$xr.log "config4 XR=$xr"
done
for xw in 10 6; do
	SSW_GPU_XR_WINDOW=$xw timeout 150 python bench.py --config 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/o_c4_xw$xw.log 2>&1; show gpurun_out/o_c4_xw$xw.log "config4 XR_WINDOW=$xw"
done
