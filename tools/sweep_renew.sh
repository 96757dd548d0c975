cd $GRAFT_REPO_ROOT
for L in default d24 d64 d96; do
  for C in C2:2.5e8 C1:1e8 C4:1e8 C8:2e6 C1_open:2e6; do
    cfg=${C%%:*}; n=${C##*:}
    if [ $L = default ]; then unset RTMC_LIBRARY; else export RTMC_LIBRARY=$PWD/build_ab/librtmc_$L.so; fi
    v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --config $cfg --photons-per-step $n | python3 -c "import sys,json; print('%.4g' % json.loads(sys.stdin.read())['value'])") || exit 1
    echo "$L $cfg $v"
  done
done
