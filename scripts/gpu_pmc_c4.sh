set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_c4
C4="--reads 10000 --read-len 10000 --ref-len 20000 --flag 0 --sub 0.01 --indel 0.0025 --mask-len 5000 --steps 1 --warmup 0 --cpu-sample 0"
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -d gpurun_out/prof_c4/pmc_sq1 -o bench -- python bench.py $C4 > gpurun_out/prof_c4/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d gpurun_out/prof_c4/pmc_sq2 -o bench -- python bench.py $C4 > gpurun_out/prof_c4/pmc_sq2.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVE_DEP_WAIT SQ_INST_LEVEL_LDS -d gpurun_out/prof_c4/pmc_sq3 -o bench -- python bench.py $C4 > gpurun_out/prof_c4/pmc_sq3.log 2>&1
ls gpurun_out/prof_c4/*/
