#!/bin/bash
# round 3, call f: upper bound of what packed targets could gain (measurement-only variant without target staging work), then the
# rocprofv3 evidence of round 3
mkdir -p gpurun_out
V=complete-striped-smith-waterman-library_amd/variants
line() { python3 -c "import sys,json; o=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(sys.argv[2], o['value'], o['phases_ms_per_step'], o['roofline']['kernel'][:24])" $1 $2; }
for i in 1 2; do
timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --also none > gpurun_out/f_c2_base$i.log 2>&1; line gpurun_out/f_c2_base$i.log c2_base$i
SSW_LIB=$V/libssw_stagefree.so timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --also none > gpurun_out/f_c2_stagefree$i.log 2>&1; line gpurun_out/f_c2_stagefree$i.log c2_stagefree$i
done
bash scripts/gpu_profile_round3.sh
mkdir -p gpurun_out/profiles_round3; ls gpurun_out/profiles_round3
