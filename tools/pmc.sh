#!/bin/bash
# Hardware-counter passes over one bench step (run on the GPU box through gpurun).
# One rocprofv3 run per counter group; kernel-trace is not combined with --pmc.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${1:-r1}
# (--also "": nothing but the one launch of the workload; the counters are summed over the trace kernel's launches)
ARGS="bench.py --steps 1 --warmup 0 --photons-per-step ${2:-1e8} --no-cpu-baseline --also= --config ${3:-C2}"
mkdir -p $OUT
run() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $name -- python3 $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
if [ -n "${4:-}" ]; then export RTMC_LIBRARY=$GRAFT_REPO_ROOT/$4; fi
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_SCA
run c FETCH_SIZE
run d WRITE_SIZE
run e SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU
ls $OUT
