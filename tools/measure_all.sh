# Measurements of every configuration in one GPU session (results under gpurun_out/r3m/)
set -e
O=gpurun_out/r3m
mkdir -p $O
timeout -k 10 600 python tools/results_table.py 1e8 > $O/results_table.txt 2>&1
( echo "== hit table with runs of coplanar rectangles"; timeout -k 10 300 python tools/strips.py 1e8 2>&1 | head -4; timeout -k 10 200 python tools/rate.py 8e8 C4;
  echo "== the same scenes with RTMC_NO_RUNS=1"; RTMC_NO_RUNS=1 timeout -k 10 300 python tools/strips.py 1e8 2>&1 | head -4; RTMC_NO_RUNS=1 timeout -k 10 200 python tools/rate.py 8e8 C4 ) > $O/runs.txt 2>&1
timeout -k 10 300 python tools/small_job.py > $O/small_jobs.txt 2>&1
timeout -k 10 600 python tools/full_size.py > $O/full_size.txt 2>&1
if [ -x tools/ubench/valu_rates ]; then timeout -k 10 120 tools/ubench/valu_rates > $O/valu_rates.txt 2>&1; fi
timeout -k 10 400 python tools/cells_ab.py 2e8 C2 C3 C4 C5 C1 C6 C9 > $O/cells_ab.txt 2>&1
timeout -k 10 400 python tools/cells_ab.py 1e8 S3 S14 stack8 C2r C4r S14r tilted > $O/cells_ab_more.txt 2>&1
cat $O/results_table.txt | tail -14
