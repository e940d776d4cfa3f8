#!/bin/bash
# What i8_mel_mfma_kernel<true, 1> reads beyond its spectrogram: FETCH_SIZE of the INT8 bench command with the exactness pass
# on (stft_exact = 2: the mixer settles the elements in doubt itself, re-reading their frames' 512 samples) and off (BN_STFT_EXACT=0 seeds the option
# at load: the mixer without the float64 settle).  The difference is the settle's audio re-reads.  Run on the GPU box from the repository root.
R=${1:-r04}
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_mel_guard_fetch -- python3 bench.py --dtype i8 --batch 4096 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline >/dev/null 2>&1
BN_STFT_EXACT=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_mel_noguard_fetch -- python3 bench.py --dtype i8 --batch 4096 --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline >/dev/null 2>&1
ls gpurun_out | grep "^${R}_mel"
