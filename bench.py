#!/usr/bin/env python3
"""Throughput benchmark of the photon-transport hot path.

    python bench.py --gpus N --steps K --warmup W

A *step* is one `rtmc_run` call: `--photons-per-step` photon ids of the
benchmark scene traced on each rank's GPU (scene tables are resident in HBM
before the timed region; the only output is the small tally block).  With
N > 1 the ranks take disjoint id ranges and the integer tallies are summed
once with an RCCL all-reduce inside the timed region.

Workload (BASELINE.json configs[1]): the Lumogen Red slab "C2" -- 5 x 5 x 0.3 cm
slab enclosed by a detector box, 460 nm source, 12 rectangles; one step is
one call of 1e9 photons per GPU, which is that configuration literally.

Ranks: one process per GPU.  Under `torch.distributed.run` (WORLD_SIZE in the
environment) this process is one rank.  Started plainly with `--gpus N`, N > 1,
it starts the N ranks itself -- as child processes, before anything in this
process has touched a GPU -- waits for them and prints rank 0's line.

Prints one JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 peak
PEAK_HBM_GBS = 8000.0


def flops_per_iteration(n_elements):
    """Algorithmic work of one loop iteration: 61 flop per element tested
    (rectangle hit 35, distance 9, side test 17) + 250 for the optical
    operation (Fresnel 105, reflect/refract 40, polarisation rotation 105)."""
    return 61 * n_elements + 250


def kernel_source_digest():
    """sha256 over the sources the trace kernel is built from: ties a counter
    measurement to the code it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("raytracer_amd/csrc/rtmc_device.h", "raytracer_amd/csrc/rtmc.hip", "include/rtmc.h"):
        with open(os.path.join(REPO, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def committed_counters(config, per_step):
    """The committed counter passes (profiles/*_traffic.json) -- only if they were
    taken on this workload, this launch size and exactly these kernel sources;
    None otherwise, so that a figure from an older kernel is never reported for
    a newer one."""
    best = None
    pdir = os.path.join(REPO, "profiles")
    digest = kernel_source_digest()
    if os.path.isdir(pdir):
        for name in sorted(os.listdir(pdir)):
            if name.endswith("_traffic.json"):
                with open(os.path.join(pdir, name)) as fh:
                    t = json.load(fh)
                if (t.get("workload") == config and int(t.get("photons_per_launch", 0)) == per_step
                        and t.get("kernel_source_sha256") == digest):
                    best = t
    return best


def measured_traffic(config, per_step):
    """HBM bytes per launch from the committed FETCH_SIZE / WRITE_SIZE passes, or None (JSON null)."""
    t = committed_counters(config, per_step)
    return None if t is None else float(t["hbm_bytes_per_launch"])


N_SIMD = 256 * 4           # SIMDs of one MI355X
# SIMD cycles one wave-instruction occupies the vector pipe, measured per instruction class at the
# kernel's own occupancy (tools/ubench/valu_rates.hip, profiles/r02_valu_rates.txt)
CYCLES_PLAIN, CYCLES_TRANS = 2.4, 8.2


def executed_work(config, per_step, kernel_s, clock_hz=2.37e9):
    """What the kernel executed per launch, from the committed counter passes (same digest rule as
    the traffic): VALU lane occupancy, the share of the SIMDs' vector-issue cycles it used, and the
    float32 operations its FMA / MUL / ADD instructions performed on active lanes.  None if the
    counters do not belong to the present sources."""
    t = committed_counters(config, per_step)
    if t is None or "SQ_INSTS_VALU" not in t:
        return None
    valu, lanes = float(t["SQ_INSTS_VALU"]), float(t["SQ_THREAD_CYCLES_VALU"])
    trans = float(t.get("SQ_INSTS_VALU_TRANS_F32", 0.0))
    occupancy = lanes / (64.0 * valu)
    # a lower bound: instructions with a scalar operand, v_cmpx and the 32x32->64 multiplies of Philox
    # cost more than the plain rate (4.2 / 4.8 / 5.4 cycles in a stream of their own)
    issue_cycles = (valu - trans) * CYCLES_PLAIN + trans * CYCLES_TRANS
    wave_flops = 2.0 * float(t["SQ_INSTS_VALU_FMA_F32"]) + float(t["SQ_INSTS_VALU_MUL_F32"]) + float(t["SQ_INSTS_VALU_ADD_F32"])
    return {
        "lane_occupancy": occupancy,
        "valu_issue_frac": issue_cycles / (N_SIMD * clock_hz * kernel_s),
        # the same with the nominal two cycles of a 64-lane instruction on a 32-lane pipe (157.3 TFLOP/s
        # = 1024 SIMDs x 32 lanes x 2 flop x 2.4 GHz) and eight for a transcendental
        "valu_issue_frac_nominal": ((valu - trans) * 2.0 + trans * 8.0) / (N_SIMD * clock_hz * kernel_s),
        "executed_fp32_tflops": wave_flops * 64.0 * occupancy / kernel_s / 1e12,
        "executed_fp32_tflops_all_lanes": wave_flops * 64.0 / kernel_s / 1e12,
        "valu_wave_instructions_per_photon": valu / per_step,
        "note": "SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU, SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F32 of one launch "
                "(rocprofv3 --pmc passes, profiles/), priced with the measured issue cost per instruction class "
                "(%.1f SIMD cycles, %.1f for transcendentals: a lower bound) at %.2f GHz; duration = this run's kernel time"
                % (CYCLES_PLAIN, CYCLES_TRANS, clock_hz * 1e-9),
    }


def launch_ranks(args, argv):
    """Start one child process per GPU and relay rank 0's JSON line.  Nothing
    here imports torch or calls into the HIP runtime: a process that has
    initialised a GPU must not be the parent of the ranks."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    text = out0.decode("utf-8", "replace")
    sys.stdout.write(text)
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        return max(1, max(abs(c) for c in codes))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--photons-per-step", type=float, default=1e9,
                    help="weak scaling (default): photons per step on every GPU")
    ap.add_argument("--total-photons", type=float, default=None,
                    help="strong scaling: photons per step over all ranks together; every rank takes "
                         "its shard_range of them")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=float, default=1.5e8)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo to rehearse N ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (rehearsal only)")
    ap.add_argument("--also", default="C3:2e9,C4:2e9,C5:4e9",
                    help="single-GPU runs only: further BASELINE configurations traced once each after the timed "
                         "region (name:photons, comma separated; empty: none); reported as other_configs")
    ap.add_argument("--dump-tallies", default=None,
                    help="rank 0 writes the summed tallies of the timed steps to this .npz (tests)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; running %d rank(s)\n"
                         % (args.gpus, world, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = local_rank if args.device is None else args.device
    # a launcher's environment (torch.distributed.run, or launch_ranks above) makes this a rank
    # even when it is the only one: the process group and the RCCL all-reduce are then exercised
    grouped = "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ
    if grouped:
        torch.cuda.set_device(device)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)

    from raytracer_amd import RayTrace, configs, _abi, distributed
    from raytracer_amd import scene as sc

    strong = args.total_photons is not None
    if strong:      # one step = `total` photon ids, cut into contiguous blocks, one per rank
        total = int(args.total_photons)
        shard_first, per_step = distributed.shard_range(total, rank, world)
    else:           # one step = per_step ids on every rank
        per_step = int(args.photons_per_step)
        total, shard_first = per_step * world, rank * per_step
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.make_job(args.config, per_step))
    scene = rt.Scene()
    n_el = scene.n_elements

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # every (rank, step) gets its own photon-id range
    def first_id(step_index):
        return step_index * total + shard_first

    block = sc.new_tallies(n_el)
    # set-up, not a step: device context, scene tables into HBM, kernel code object
    # (a call for zero photons prepares the device and launches nothing)
    scene.run(0, seed=args.seed, first_photon=0, device=device, tallies=sc.new_tallies(n_el))
    for w in range(args.warmup):
        scene.run(per_step, seed=args.seed, first_photon=first_id(w), device=device,
                  tallies=sc.new_tallies(n_el))
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        scene.run(per_step, seed=args.seed, first_photon=first_id(args.warmup + k), device=device,
                  tallies=block)
        kernel_ms.append(scene.last_kernel_ms(device))
    distributed.allreduce_tallies(block, device)
    barrier()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tall = sc.tallies_to_dict(*block)

    out = None
    if rank == 0 and args.dump_tallies:
        np.savez(args.dump_tallies, first_photon=first_id(args.warmup) - shard_first,
                 n_requested=total * args.steps,
                 **{k: tall[k] for k in list(_abi.TALLY_SCALARS) + list(_abi.TALLY_ARRAYS) + ["spectrum"]})
    if rank == 0:
        total_photons = total * args.steps
        assert tall["n_photons"] == total_photons, (tall["n_photons"], total_photons)
        iters_per_launch = tall["n_steps"] / (args.steps * world)
        avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
        flops = iters_per_launch * flops_per_iteration(n_el)
        achieved = flops / avg_kernel_s / 1e12
        traffic = measured_traffic(args.config, per_step)
        edges = [2, 3, 4, 5]
        out = {
            "metric": "photons/sec",
            "value": total_photons / elapsed,
            "unit": "photons/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: Lumogen Red slab 5x5x0.3 cm in a detector box, 460 nm "
                                   "source, %d rectangles" % (args.config, n_el)
                       if args.config == "C2" else args.config,
                       "photons_per_step_per_gpu": per_step if not strong else None,
                       "photons_per_step_total": total,
                       "parallelism": "photon-id ranges, %d rank(s)" % world,
                       "iterations_per_photon": tall["n_steps"] / tall["n_photons"],
                       "optical_efficiency": float(tall["counter"][edges].sum()) / tall["n_photons"],
                       "internal_converted": tall["n_internal_converted"] / tall["n_photons"],
                       "capped": int(tall["n_capped"]), "lost": int(tall["n_lost"])},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                         "traffic": traffic,
                         "kernel": ("trace_kernel<float, FEAT_ALL>" if scene.info()["all_features_kernel"]
                                    else "trace_kernel<float, %d>" % scene.cells()["kernel_features"]),
                         "closed_boxes_in_use": scene.cells()["in_use"],
                         "kernel_ms_avg": avg_kernel_s * 1e3,
                         "flop_per_iteration": flops_per_iteration(n_el),
                         "iterations_per_launch": iters_per_launch,
                         "executed": executed_work(args.config, per_step, avg_kernel_s),
                         # the photons never leave the registers: what crosses HBM is the measured
                         # traffic (scene tables in, per-workgroup tallies out), not 96 B per photon
                         "hbm": {"achieved": None if traffic is None else traffic / avg_kernel_s / 1e9,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": None if traffic is None else traffic / avg_kernel_s / 1e9 / PEAK_HBM_GBS,
                                 "photon_record_bytes_kept_in_registers": 96}},
            "collective": distributed.collective_info(block),
        }
        if world == 1 and args.also and args.config == "C2" and not strong:
            out["other_configs"] = other_configs(args.also, args.seed, device)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(rt, int(args.cpu_sample), args.seed, args.config)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return out


def other_configs(spec, seed, device):
    """The other BASELINE configurations (co-doped slab with donor -> acceptor transfer, stacked slabs with
    reflective edges, sun-spectrum source) traced once each on this GPU, outside the timed region of the
    headline workload: photons/s by the kernel time of one launch, with what identifies the run."""
    from raytracer_amd import RayTrace, configs
    result = {}
    for item in spec.split(","):
        name, _, count = item.partition(":")
        n = int(float(count or 1e9))
        rt = RayTrace.RayTrace(demo_scene=False)
        rt.LoadJob(configs.make_job(name, n))
        s = rt.Scene()
        try:
            s.run(1000000, seed=seed + 1, device=device)                  # scene upload and code object
            t0 = time.perf_counter()
            t = s.run(n, seed=seed, device=device)
            wall = time.perf_counter() - t0
            edges = [2, 8, 14] if name == "C4" else [2, 3, 4, 5]       # C4: the one detector edge of each of its slabs
            result[name] = {"photons": n, "photons_per_s": n / (s.last_kernel_ms(device) * 1e-3),
                            "photons_per_s_wall": n / wall, "kernel_ms": s.last_kernel_ms(device),
                            "iterations_per_photon": t["n_steps"] / t["n_photons"],
                            "optical_efficiency": float(t["counter"][edges].sum()) / t["n_photons"],
                            "every_photon_has_a_fate": bool(t["n_photons"] == n),
                            "closed_boxes_in_use": s.cells()["in_use"]}
        finally:
            s.close()
    return result


def cpu_baseline(rt, sample, seed, config):
    """The float64 CPU checker (oracle/) on a bounded sample of the same workload."""
    from oracle import oracle
    threads = min(16, os.cpu_count() or 1)
    o = oracle.OracleScene.from_raytrace(rt)
    t0 = time.perf_counter()
    r = o.run_threads(sample, seed=seed, threads=threads)
    dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "photons/s", "cores": threads, "kind": "port",
           "sample": "%d photons of the same scene, float64 scalar C restatement of the "
                     "reference loop, %d host threads, %.1f s" % (sample, threads, dt),
           "optical_efficiency": float(r["counter"][[2, 3, 4, 5]].sum()) / r["n_photons"]}
    ref = reference_python_rate(config)
    if ref:
        out["reference_python"] = ref
    return out


def reference_python_rate(config):
    """Photons/s of the reference's own Python loop on one core, from the wall
    times stored with its seeded runs (tests/golden/ref_tallies.json; measured
    in the build container, the reference cannot travel to the GPU box)."""
    path = os.path.join(REPO, "tests", "golden", "ref_tallies.json")
    try:
        with open(path) as fh:
            runs = [r for r in json.load(fh) if r.get("config") == config and r.get("seconds")]
    except OSError:
        return None
    if not runs:
        return None
    n, secs = sum(r["n"] for r in runs), sum(r["seconds"] for r in runs)
    return {"value": n / secs, "unit": "photons/s", "cores": 1, "kind": "reference",
            "sample": "%d photons in %d seeded runs of the reference's RayTrace.runSimulation "
                      "(Python, one core, build container), %.0f s" % (n, len(runs), secs)}


if __name__ == "__main__":
    main()
