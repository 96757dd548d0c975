"""Copies the evidence of one tools/profile_round.sh session from gpurun_out/prof_<tag>/ into
profiles/<tag>_*: the bench lines, the rocprofv3 kernel statistics, the counter summary, the
traffic file (whose digest must match the present kernel sources) and the section report without
its warm-up launches.  usage: python tools/install_profiles.py r02   (in the build container)"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
for a, b in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
             ("stats/bench_kernel_stats.csv", "bench_kernel_stats.csv"), ("pmc_summary.txt", "pmc_summary.txt"),
             ("traffic.json", "traffic.json"), ("bench_with_traffic.json", "bench_with_traffic.json")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, "%s_%s" % (tag, b)))
out, lines, i = [], open(os.path.join(src, "sections.txt")).read().split("\n"), 0
while i < len(lines):
    if lines[i].startswith("-- "):          # a warm-up launch: skip its report
        i += 1
        while i < len(lines) and not lines[i].startswith("== "):
            i += 1
        continue
    out.append(lines[i])
    i += 1
open(os.path.join(dst, tag + "_sections.txt"), "w").write("\n".join(out))
stamp = json.load(open(os.path.join(dst, tag + "_traffic.json")))["kernel_source_sha256"]
ok = stamp == bench.kernel_source_digest()
b = json.load(open(os.path.join(dst, tag + "_bench.json")))
print("traffic digest %s the present sources; bench %.4g photons/s, %.2f ms per step, kernel %.2f ms, frac %.3f"
      % ("matches" if ok else "DOES NOT MATCH", b["value"], b["ms_per_step"], b["roofline"]["kernel_ms_avg"], b["roofline"]["frac"]))
sys.exit(0 if ok else 1)
