#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ for both bench configurations (run on the GPU box through gpurun):
# kernel-trace stats, then separate --pmc passes (FETCH_SIZE; WRITE_SIZE + L2 hit/miss; SQ instruction mix; SQ wait/busy breakdown).
# The program follows `--` directly (python3 bench.py ...): no env / shell hop under the profiler.
R=${1:-r03}
for d in i8 f32; do
  X="--dtype f32 --batch 1024"; [ $d = i8 ] && X="--dtype i8 --batch 4096"   # explicit batch: no second configuration in the trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_$d -- python3 bench.py $X --steps 20 --repeats 5 --no-cpu-baseline > gpurun_out/${R}_$d.json 2>/dev/null
  echo "$d stats done"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_${d}_fetch -- python3 bench.py $X --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline >/dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${R}_${d}_write -- python3 bench.py $X --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline >/dev/null 2>&1
  echo "$d traffic done"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/${R}_${d}_sq -- python3 bench.py $X --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline >/dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${R}_${d}_sq2 -- python3 bench.py $X --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline >/dev/null 2>&1
  echo "$d sq done"
done
ls gpurun_out | grep "^${R}_" | head -30
