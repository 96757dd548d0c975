"""One loop iteration in float32 from every (photon, iteration) state of the golden traces:
stores the float32 results per configuration for analysis off the GPU box (the float64
states they are compared with come from the CPU checker and are recomputed there).
usage (GPU box): python tools/f32_step_dump.py gpurun_out/r3/step_f32.npz"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import helpers as H  # noqa: E402
from oracle import oracle  # noqa: E402
from raytracer_amd import _abi  # noqa: E402

oracle.build()
out = {}
for name in ["C1", "C2", "C3", "C4", "C5", "C6", "C7", "C8", "C9", "C2r"]:
    g = H.load_trace(name)
    rt = H.raytrace_for(name)
    s = rt.Scene()
    states, picks, big, tape, off = H.single_iteration_inputs(oracle, rt, g, 40)
    s.step_tape(big, tape, off, 1, launch=False, precision=_abi.F32)
    for key in ("loc", "dir", "efield", "dist_left", "status", "material", "last_element", "reflected",
                "tape_pos", "wavelength"):
        out["%s_%s" % (name, key)] = big[key]
    s.close()
np.savez(sys.argv[1], **out)
print("stored", sys.argv[1])
