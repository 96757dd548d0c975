set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r01
timeout -k 10 300 python bench.py > gpurun_out/prof_r01/bench.json 2> gpurun_out/prof_r01/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01/stats -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_r01/bench_under_rocprof.json 2> gpurun_out/prof_r01/rocprof.err
bash tools/pmc.sh r01 2.5e8 C2 > gpurun_out/prof_r01/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_r01 > gpurun_out/prof_r01/pmc_summary.txt
ls gpurun_out/prof_r01/stats | head
