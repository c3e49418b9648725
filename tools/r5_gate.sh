# round 5 experiments on the worker pool's GPU sharing: bash tools/r5_gate.sh
for q in 1 2 3; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python tools/map_pool_rate.py 512 8 2>/dev/null | grep "^workers\|GPU busy" | cut -c1-100
done
echo "== default"
python tools/map_pool_rate.py 512 8 2>/dev/null | grep "^workers\|GPU busy" | cut -c1-100
echo "== HSA_ENABLE_SDMA=0"
HSA_ENABLE_SDMA=0 python tools/map_pool_rate.py 512 8 2>/dev/null | grep "^workers\|GPU busy" | cut -c1-100
