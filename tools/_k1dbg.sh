cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 8 15; do
  export BN_GUARD_DBG=$d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k1dbg_$d -- python3 bench.py --dtype i8 --batch 4096 --steps 10 --no-cpu-baseline > /dev/null 2>&1
  f=$(ls gpurun_out/k1dbg_$d/*/*kernel_stats.csv | head -1)
  echo "dbg=$d $(grep StftGuard $f | grep MelOut | awk -F, '{print $(NF-4)}')"
done
