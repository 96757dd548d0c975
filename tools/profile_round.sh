# One GPU session of evidence for a round: tools/profile_round.sh r02
#   bench.py (default arguments), rocprofv3 --kernel-trace --stats of the same command, the --pmc
#   passes of pmc.sh over one launch of the bench's size, their summary, the traffic file that
#   bench.py reports as roofline.traffic (stamped with the digest of the kernel sources), and the
#   cycle stamps of the diagnostic build.  Results under gpurun_out/prof_<round>/; copy into profiles/.
set -e
R=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_$R
mkdir -p $O
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
bash tools/pmc.sh $R 1e9 C2 > $O/pmc.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$R > $O/pmc_summary.txt
python3 tools/traffic_json.py $O/pmc_summary.txt 1000000000 C2 > $O/traffic.json
timeout -k 10 300 python3 tools/sections.py 1e8 C2 C3 C4 C2r C1 2> $O/sections.txt
# the bench line once more, now that this session's traffic file is in place (on this box's copy of the tree)
cp $O/traffic.json profiles/${R}_traffic.json
timeout -k 10 400 python3 bench.py > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err
find $O/stats -name "*kernel_stats.csv" | head -3
cat $O/bench.json
