#!/bin/bash
# Counters of the fused INT8 tail kernel (either form; BN_I8_TAIL_MFDW=0 in the environment selects i8_tail_kernel): three rocprofv3 --pmc passes
# of the INT8 bench command, summarised per kernel by tools/pmc_summary.py.  Run on the GPU box from the repository root.
R=${1:-r05_tail}
X="--dtype i8 --batch 4096 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/${R}_a -- python3 bench.py $X >/dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${R}_b -- python3 bench.py $X >/dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS --output-format csv -d gpurun_out/${R}_c -- python3 bench.py $X >/dev/null 2>&1
python3 tools/pmc_summary.py "gpurun_out/${R}_a/**/*counter_collection.csv" "gpurun_out/${R}_b/**/*counter_collection.csv" "gpurun_out/${R}_c/**/*counter_collection.csv" > gpurun_out/${R}_summary.txt
grep -A1 "i8_tail" gpurun_out/${R}_summary.txt
