# A/B of guard variants: throughput, photons that end elsewhere than in float64, seam photons
for lib in "$@"; do
  RTMC_LIBRARY=$PWD/ab/$lib.so timeout -k 10 200 python tools/f32_moved.py 5e6 C1 C2 C3 C4 C8 C2r || exit 1
  RTMC_LIBRARY=$PWD/ab/$lib.so timeout -k 10 200 python -u -m pytest tests/test_gpu_seams.py -m gpu -q -s -k "sit_on_a_seam" --timeout 100 --timeout-method thread 2>&1 | grep "seam,.*f32"
done
bash tools/ab.sh 2e8 "C2 C3 C1" $(for l in "$@"; do echo ab/$l.so; done)
