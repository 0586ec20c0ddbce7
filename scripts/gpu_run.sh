#!/bin/bash
# GPU-box calls of round 4 and later, as named steps:   gpurun -- 'bash scripts/gpu_run.sh <step> [<step> ...]'
# Outputs land in gpurun_out/<tag>_*.log (tag = $SSW_RUN_TAG, default "r4"); what is evidence gets copied to profiles/ by hand or by
# scripts/summarize_profile.py.  Every step runs under its own timeout so that a hang costs one step, not the call.
#   tests          the whole GPU test-suite
#   bench          the default bench line exactly as the driver runs it (config 2 + `also`)
#   bench:<name>:<args...>   any other bench line, e.g. bench:c6:--config\ 6  (args may not contain spaces other than separators: use '+')
#   xbench:<name>:<ENV=V,...>:<args...>   the same with environment variables (hooks library, variant builds: SSW_LIB=@/complete-.../variants/libssw_x.so)
#   fuzz           scripts/gpu_fuzz.py (batch ABI, empties, wide alphabets) and scripts/abi_fuzz.py (single-pair regime), two seeds each
#   config6        the README's benchmark shape: default / serial buckets / -m1 -x3 -o5 -e2 / flag 2
#   overlap4       config 4 alone and through two pool workers on the one GPU (does the tail of one half hide under the fill of the other?)
#   dbx            begin positions / CIGARs against a whole database in one call vs score only vs the per-target loop (scripts/gpu_dbx_bench.py)
#   latency        one ssw_align call of the drop-in ABI (scripts/gpu_latency.py)
#   literal        the lane-model kernel (gapO <= gapE) at 2 000 and 20 000 reads
#   dropin         the reference main.c on libssw.so (one ssw_align per read) beside ssw_test_gpu, 10 000 reads x 1 Mb (scripts/gpu_dropin_cli.py)
#   profile        rocprofv3 kernel-trace + PMC passes (scripts/gpu_profile.sh)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TAG=${SSW_RUN_TAG:-r4}
short() { python3 - "$1" "$2" <<'PY'
import sys, json
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], o.get('value'), 'n_gpus', o.get('n_gpus'), o.get('phases_ms_per_step'), 'roofline', (o.get('roofline') or {}).get('frac'),
          'parity', (o.get('parity') or {}).get('mismatching_alignments'), 'cpu', (o.get('cpu_baseline') or {}).get('value'))
    for k, v in (o.get('also') or {}).items():
        if isinstance(v, dict):
            print('   also', k, v.get('value'), v.get('phases_ms_per_step'), 'roofline', v.get('roofline_frac'), 'parity', v.get('parity'), v.get('error'))
except Exception as e:
    print(sys.argv[2], 'no JSON line:', e)
    print(open(sys.argv[1]).read()[-1500:])
PY
}
bench() { local name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/${TAG}_bench_$name.log 2>&1; short gpurun_out/${TAG}_bench_$name.log $name; }
for step in "$@"; do
  case "$step" in
    tests) timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log ;;
    bench) bench default --gpus 1 --steps 20 --warmup 5 ;;      # (the driver's command line: python3 bench.py --gpus 1 --steps 20 --warmup 5)
    bench:*) IFS=: read -r _ name args <<< "$step"; bench "$name" ${args//+/ } ;;
    xbench:*) IFS=: read -r _ name envs args <<< "$step"; ( for kv in ${envs//,/ }; do export "${kv//@/$PWD}"; done; bench "$name" ${args//+/ } ) ;;      # xbench:<name>:<ENV=V,ENV=V>:<args>  (a path in V may use @ for $PWD)
    fuzz) for s in ${SSW_FUZZ_SEEDS:-1 2}; do timeout $(( ${SSW_FUZZ_SECS:-150} + 250 )) python scripts/gpu_fuzz.py ${SSW_FUZZ_SECS:-150} $s 2>/dev/null | tee -a gpurun_out/${TAG}_gpu_fuzz.json | cut -c1-600; done
          for s in ${SSW_FUZZ_SEEDS:-1 2}; do timeout $(( ${SSW_FUZZ_SECS:-150} + 150 )) python scripts/abi_fuzz.py ${SSW_FUZZ_SECS:-150} $s 2>/dev/null | tee -a gpurun_out/${TAG}_abi_fuzz.json | cut -c1-600; done ;;
    config6)
      bench c6 --config 6
      SSW_LIB=$PWD/complete-striped-smith-waterman-library_amd/libssw_hooks.so SSW_GPU_SERIAL_BUCKETS=1 bench c6_serial_buckets --config 6 --cpu-sample 0
      bench c6_m1x3o5e2 --config 6 --match 1 --mismatch 3 --gap-open 5 --gap-extend 2
      bench c6_flag2 --config 6 --flag 2 --cpu-sample 0 ;;
    overlap4)
      bench c4 --config 4 --steps 2 --warmup 1 --cpu-sample 0
      bench c4_pool2 --config 4 --pool 2 --steps 2 --warmup 1 --cpu-sample 0 ;;
    dbx) timeout 900 python scripts/gpu_dbx_bench.py > gpurun_out/${TAG}_dbx.log 2>&1; tail -c 3000 gpurun_out/${TAG}_dbx.log ;;
    latency) timeout 300 python scripts/gpu_latency.py > gpurun_out/${TAG}_latency.log 2>&1; tail -c 2000 gpurun_out/${TAG}_latency.log ;;
    literal)
      bench literal_2k --reads 2000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 --also none
      bench literal_20k --reads 20000 --gap-open 1 --gap-extend 1 --steps 1 --warmup 1 --cpu-sample 200 --also none ;;
    dropin) timeout 600 python scripts/gpu_dropin_cli.py > gpurun_out/${TAG}_dropin_cli.log 2>&1; tail -c 2500 gpurun_out/${TAG}_dropin_cli.log ;;
    profile) bash scripts/gpu_profile.sh ${SSW_PROFILE_ROUND:-round6} > gpurun_out/${TAG}_profile.log 2>&1; tail -6 gpurun_out/${TAG}_profile.log ;;
    timeline:*) IFS=: read -r _ name args <<< "$step"; bash scripts/gpu_trace_timeline.sh "$name" ${args//+/ } > gpurun_out/${TAG}_timeline_$name.log 2>&1; tail -3 gpurun_out/${TAG}_timeline_$name.log ;;
    *) echo "unknown step $step" ;;
  esac
done
du -sh gpurun_out
