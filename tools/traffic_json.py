"""HBM bytes per launch of the trace kernel from the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc.sh (as summarised by pmc_summary.py), written in the form bench.py reads: stamped
with the digest of the kernel sources the counters were taken on.
usage: python tools/traffic_json.py pmc_summary.txt photons_per_launch workload"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
vals = {}
for line in open(sys.argv[1]):
    parts = line.split()
    if len(parts) == 2:
        vals[parts[0]] = float(parts[1])
# rocprofv3 reports both in KB; FETCH_SIZE is doubled (gfx950: the counter counts 64-byte
# requests as 32-byte ones for reads)
fetch_kb, write_kb = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
counters = {k: vals[k] for k in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32",
                                 "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_SALU",
                                 "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES") if k in vals}
print(json.dumps({
    **counters,
    "workload": sys.argv[3], "photons_per_launch": int(sys.argv[2]), "kernel": "trace_kernel<float, …> of the workload (see the bench line)",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "kernel_source_sha256": bench.kernel_source_digest(),
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc.sh via "
            "tools/profile_round.sh); FETCH_SIZE doubled as the gfx950 correction for reads prescribes; values are KB",
}, indent=1))
