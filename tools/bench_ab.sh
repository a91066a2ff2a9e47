#!/bin/bash
# A/B the scan kernels on the GPU box: prints one compact line per variant.
# usage: tools/bench_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each arg = one env setting string)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "$@"; do
  env $cfg python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1 > /tmp/_b.json
  python - "$cfg" <<'PY'
import json,sys
d=json.load(open('/tmp/_b.json')); r=d['roofline']
print("%-40s q/s=%9.0f ms/step=%7.3f kernel_ms=%7.3f TF=%7.1f (%.3f) HBM=%6.0f GB/s unproven=%d" % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel_ms'], r['mfma_tflops'], r['mfma_frac'], r['hbm_gbps'], d['config']['unproven_queries']))
PY
done
