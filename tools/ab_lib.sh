#!/bin/bash
# A/B of two BUILDS of libbirdnet_hip.so inside one GPU session (box-to-box differences of 3-7 % hide anything smaller): the headline bench, three
# interleaved rounds; prints ms per step and the per-operator times of the layers named after the two libraries.
#   bash tools/ab_lib.sh <old.so> <new.so> [layer ...]        e.g.  bash tools/ab_lib.sh lib_old.so lib_new.so t107 t110
# (BIRDNET_HIP_LIB selects the library the Python binding loads.)
OLD=$(realpath "$1"); NEW=$(realpath "$2"); shift 2
LAYERS="${*:-stft tail}"
for r in 1 2 3; do
  for v in "$OLD" "$NEW"; do
    BIRDNET_HIP_LIB=$v python bench.py --no-extras --no-cpu-baseline --repeats 10 2>/dev/null | LAYERS="$LAYERS" python -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
st = {s['layer']: s['avg_ms'] for s in d['stages']}
print(os.path.basename('$v'), d['ms_per_step'], *[(l, st.get(l)) for l in os.environ['LAYERS'].split()])"
  done
done
