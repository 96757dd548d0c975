"""One rank of a sharded `RayTrace.runSimulation` (started by the tests).

    python dist_worker.py CONFIG N_PHOTONS SEED OUT.npz [fake]

Reads RANK / WORLD_SIZE / MASTER_* from the environment, joins a gloo group,
runs the simulation (every rank on GPU 0, or with `fake` on no GPU at all: a
stand-in scene whose tallies are a known function of the photon ids), and
rank 0 writes the all-reduced tallies.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class FakeScene:
    """Stands in for raytracer_amd.scene.Scene: photon id i ends on element
    i % n_elements after (i % 7) + 1 iterations."""

    def __init__(self, optical_elements, photon, material_dict=None, max_iterations=0):
        self.n_elements = len(optical_elements)
        self.calls = []

    def run(self, n_photons, seed=0, first_photon=0, device=0, precision=0, tallies=None):
        from raytracer_amd import scene as sc
        t, arrays = tallies
        ids = np.arange(first_photon, first_photon + n_photons, dtype=np.uint64)
        el = (ids % np.uint64(self.n_elements)).astype(np.int64)
        arrays["counter"] += np.bincount(el, minlength=self.n_elements).astype(np.uint64)
        arrays["energy_fx"] += np.bincount(el, weights=(ids % np.uint64(1000)).astype(np.float64),
                                           minlength=self.n_elements).astype(np.uint64)
        spec = arrays["spectrum"].reshape(self.n_elements, -1)
        np.add.at(spec, (el, (ids % np.uint64(3800)).astype(np.int64)), np.uint64(1))
        t.n_photons += int(n_photons)
        t.n_steps += int(((ids % np.uint64(7)) + np.uint64(1)).sum())
        t.n_dumped += int(n_photons)
        self.calls.append((int(first_photon), int(n_photons), int(seed)))
        return sc.tallies_to_dict(t, arrays)


def main():
    config, n, seed, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    fake = len(sys.argv) > 5 and sys.argv[5] == "fake"
    import torch.distributed as dist
    from raytracer_amd import RayTrace, _abi, configs
    from raytracer_amd import scene as sc
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    if fake:
        sc.Scene = FakeScene
    rt = RayTrace.RayTrace(demo_scene=False)
    rt.LoadJob(configs.make_job(config, n))
    rt.Seed = seed if seed >= 0 else None
    rt.Device = 0
    t = rt.runSimulation()
    rank = int(os.environ.get("RANK", "0"))
    if fake:
        first, count = rt.Scene().calls[0][:2] if rt.Scene().calls else (0, 0)
        np.savez(out + ".rank%d.npz" % rank, first=first, count=count, seed=rt.LastSeed)
    if rank == 0:
        np.savez(out, seed=rt.LastSeed,
                 **{k: t[k] for k in list(_abi.TALLY_SCALARS) + list(_abi.TALLY_ARRAYS) + ["spectrum"]})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
