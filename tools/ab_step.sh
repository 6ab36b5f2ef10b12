#!/bin/bash
# A/B of two builds of the library on ONE box: bash tools/ab_step.sh <other.so> [rounds] -> ms/step and GEMM us/launch, alternating
set -e -o pipefail
OTHER=$1; N=${2:-2}
for i in $(seq $N); do
  MFNET_HIP_LIB=$OTHER timeout -k 10 150 python bench.py --no-cpu --no-h2d --steps 60 > gpurun_out/ab_a.json
  timeout -k 10 150 python bench.py --no-cpu --no-h2d --steps 60 > gpurun_out/ab_b.json
  python - <<'PY'
import json
for tag, f in (("other", "gpurun_out/ab_a.json"), ("tree ", "gpurun_out/ab_b.json")):
    d = json.load(open(f))
    print(tag, "ms/step", d["ms_per_step"], " gemm_big us", d["roofline"]["avg_launch_us"], "frac", d["roofline"]["frac"], " all-gemm ms", d["roofline_all_gemm"]["gemm_ms_per_step"], flush=True)
PY
done
