#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ for both bench configurations (run on the GPU box through gpurun):
# kernel-trace stats, then three separate --pmc passes (FETCH_SIZE; WRITE_SIZE + L2 hit/miss; SQ instruction mix).
for d in f32 i8; do
  X="--batch 1024"; [ $d = i8 ] && X="--dtype i8 --batch 4096"   # explicit batch: no second configuration in the trace
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p4_$d -- python3 bench.py $X --steps 20 --no-cpu-baseline > gpurun_out/p4_$d.json 2>/dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p4_${d}_fetch -- python3 bench.py $X --steps 3 --warmup 1 --no-cpu-baseline >/dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/p4_${d}_write -- python3 bench.py $X --steps 3 --warmup 1 --no-cpu-baseline >/dev/null 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/p4_${d}_sq -- python3 bench.py $X --steps 3 --warmup 1 --no-cpu-baseline >/dev/null 2>&1
done
ls gpurun_out | head -30
