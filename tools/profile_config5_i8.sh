#!/bin/bash
# rocprofv3 evidence for BASELINE configs[4] in INT8 (own exporter, hybrid frontend with max normalisation + PCEN, 1024 chunks of 3 s): kernel-trace stats,
# then separate --pmc passes, like tools/profile_all.sh.  Run on the GPU box from the repository root; the program follows `--` directly.
R=${1:-r04}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_c5i8 -- python3 tools/config5_i8_bench.py 1024 12 hybrid > gpurun_out/${R}_c5i8.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_c5i8_fetch -- python3 tools/config5_i8_bench.py 1024 2 hybrid >/dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${R}_c5i8_write -- python3 tools/config5_i8_bench.py 1024 2 hybrid >/dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/${R}_c5i8_sq -- python3 tools/config5_i8_bench.py 1024 2 hybrid >/dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${R}_c5i8_sq2 -- python3 tools/config5_i8_bench.py 1024 2 hybrid >/dev/null 2>&1
ls gpurun_out | grep "^${R}_c5i8"
