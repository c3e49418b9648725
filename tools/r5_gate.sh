# round 5 experiments on the worker pool's GPU sharing: bash tools/r5_gate.sh
for q in 1 2; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python tools/map_pool_rate.py 512 8 12 2>/dev/null | grep workers | cut -c1-90
done
echo "== HSA_ENABLE_SDMA=0"
HSA_ENABLE_SDMA=0 python tools/map_pool_rate.py 512 8 12 2>/dev/null | grep workers | cut -c1-90
echo "== 6 / 7 workers"
python tools/map_pool_rate.py 512 6 7 2>/dev/null | grep workers | cut -c1-90
