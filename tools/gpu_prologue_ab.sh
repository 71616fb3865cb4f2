#!/bin/bash
# Same-box A/B of the "first frame before the tables" prologue (ab/libcrnsense_{old,new}.so, ab/libcrnsense_ab_{old,new}.so):
# the headline batch (8.75 GiB), SURVEY §8(d) cfgH's 2 GiB batch on one stream, and the first-epoch penalty measured from
# inside the kernel (tools/gpu_wg_placement.py, trace variant 17 of the A/B build).  Interleaved: boxes and minutes differ.
O=gpurun_out/prologue_ab
mkdir -p $O
for rep in 1 2 3 4; do
  for lib in old new; do
    CRN_SENSE_LIB=$PWD/ab/libcrnsense_$lib.so timeout 300 python bench.py --steps 60 --warmup 20 --cpu-epochs 0 --no-alt --no-live-traffic > $O/headline_${lib}_$rep.json 2> $O/headline_${lib}_$rep.err
    CRN_SENSE_LIB=$PWD/ab/libcrnsense_$lib.so timeout 300 python bench.py --steps 200 --warmup 50 --epochs 6553 --cpu-epochs 0 --no-alt --no-live-traffic > $O/cfgH2g_${lib}_$rep.json 2> $O/cfgH2g_${lib}_$rep.err
    CRN_SENSE_LIB=$PWD/ab/libcrnsense_$lib.so timeout 300 python bench.py --fft 1024 --steps 60 --warmup 20 --cpu-epochs 0 --no-alt --no-live-traffic > $O/e1024_${lib}_$rep.json 2> $O/e1024_${lib}_$rep.err
    CRN_SENSE_LIB=$PWD/ab/libcrnsense_$lib.so timeout 300 python bench.py --mode ref --steps 60 --warmup 20 --cpu-epochs 0 --no-alt --no-live-traffic > $O/ref512_${lib}_$rep.json 2> $O/ref512_${lib}_$rep.err
  done
done
for lib in old new; do
  CRN_SENSE_LIB=$PWD/ab/libcrnsense_ab_$lib.so timeout 300 python tools/gpu_wg_placement.py > $O/placement_$lib.txt 2>&1
done
python - <<'PY'
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/prologue_ab/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        tag = f.split('/')[-1][:-5].rsplit('_', 1)[0]
        acc[tag].append(r['frac'])
    except Exception as e:
        print(f, "ERR", open(f.replace('.json', '.err')).read()[-300:])
for tag, v in sorted(acc.items()):
    print("%-20s frac per rep: %s  mean %.4f" % (tag, " ".join("%.4f" % x for x in v), sum(v) / len(v)))
PY
for lib in old new; do echo "== placement $lib"; cat $O/placement_$lib.txt; done
