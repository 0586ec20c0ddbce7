# rocprofv3 evidence, end of round 1 (written under gpurun_out/, summaries copied to profiles/ by scripts/summarize_profiles.py)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/prof4 gpurun_out/prof5
mkdir -p gpurun_out/prof gpurun_out/prof4 gpurun_out/prof5
python bench.py --steps 2 --warmup 1 > gpurun_out/bench_config2.log 2>&1
BENCH="python bench.py --steps 1 --warmup 1 --cpu-sample 0"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o bench -- $BENCH > gpurun_out/prof/trace.log 2>&1
SMALL="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --reads 16000"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d gpurun_out/prof/pmc_sq1 -o bench -- $SMALL > gpurun_out/prof/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d gpurun_out/prof/pmc_sq2 -o bench -- $SMALL > gpurun_out/prof/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pmc_fetch -o bench -- $SMALL > gpurun_out/prof/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pmc_write -o bench -- $SMALL > gpurun_out/prof/pmc_write.log 2>&1
C4="--reads 10000 --read-len 10000 --ref-len 100000 --flag 2 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0"
python bench.py $C4 --cpu-sample 256 > gpurun_out/bench_config4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof4/trace -o bench -- python bench.py $C4 --cpu-sample 0 > gpurun_out/prof4/trace.log 2>&1
C4S="--reads 10000 --read-len 10000 --ref-len 20000 --flag 0 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 0"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d gpurun_out/prof4/pmc_sq1 -o bench -- python bench.py $C4S > gpurun_out/prof4/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof4/pmc_fetch -o bench -- python bench.py $C4S > gpurun_out/prof4/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof4/pmc_write -o bench -- python bench.py $C4S > gpurun_out/prof4/pmc_write.log 2>&1
C5="--reads 8192 --db-targets 2048 --steps 1 --warmup 0"
python bench.py $C5 --cpu-sample 64 > gpurun_out/bench_config5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5/trace -o bench -- python bench.py $C5 --cpu-sample 0 > gpurun_out/prof5/trace.log 2>&1
du -sh gpurun_out
