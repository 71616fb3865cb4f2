#!/bin/bash
# Interleaved A/B of kernel variants of the A/B build on one box: VARIANTS="0 28" [EXTRA="--mode welch"] [REPS=4] bash tools/gpu_variant_ab.sh
export CRN_SENSE_AB=1
O=gpurun_out/${TAG:-variant_ab}
mkdir -p $O
for rep in $(seq 1 ${REPS:-4}); do
  for v in ${VARIANTS:-0 28}; do
    timeout 300 python bench.py --steps ${STEPS:-60} --warmup 20 --cpu-epochs 0 --no-alt --no-live-traffic --variant $v ${EXTRA:-} > $O/v${v}_$rep.json 2> $O/v${v}_$rep.err
  done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('$O/v*_*.json')):
    try:
        acc[f.split('/')[-1].split('_')[0]].append(json.load(open(f))['roofline']['frac'])
    except Exception:
        print(f, "ERR", open(f.replace('.json', '.err')).read()[-300:])
for v, x in sorted(acc.items(), key=lambda kv: int(kv[0][1:])):
    print("variant %-4s ${EXTRA:-} frac %s  mean %.4f" % (v[1:], " ".join("%.4f" % y for y in x), sum(x) / len(x)))
PY
