# hit table with and without runs of coplanar rectangles: throughput of the scenes that have them
for v in 0 1; do
  if [ $v = 1 ]; then export RTMC_NO_RUNS=1; echo "-- without runs"; else echo "-- with runs"; fi
  timeout -k 10 200 python tools/rate.py 8e8 C4 || exit 1
  timeout -k 10 200 python tools/rate.py 2e8 C2 || exit 1
  timeout -k 10 300 python tools/strips.py 1e8 2>&1 | grep -v "histogram in HBM" | head -8
done
