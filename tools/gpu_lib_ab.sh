#!/bin/bash
# Same-box A/B of two builds of the library (ab/libcrnsense_old.so vs ab/libcrnsense_new.so, selected with $CRN_SENSE_LIB): parity of
# the new build first (the GPU parity tests through it), then interleaved bench lines of every kernel family.  Boxes and minutes
# differ by a few %: only numbers of one call compare.   REPS=4 TAG=lib_ab bash tools/gpu_lib_ab.sh
O=gpurun_out/${TAG:-lib_ab}
mkdir -p $O
CRN_SENSE_LIB=$PWD/ab/libcrnsense_new.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_fft.py tests/test_decision_band.py -m gpu -q -x --deselect tests/test_gpu_parity.py::test_kernel_variants_agree > $O/parity_new.log 2>&1; echo "parity exit $?" >> $O/parity_new.log
tail -3 $O/parity_new.log
for rep in $(seq 1 ${REPS:-4}); do
  for lib in old new; do
    b() { local tag=$1; shift; CRN_SENSE_LIB=$PWD/ab/libcrnsense_$lib.so timeout 300 python bench.py --cpu-epochs 0 --no-alt --no-live-traffic "$@" > $O/${tag}_${lib}_$rep.json 2> $O/${tag}_${lib}_$rep.err; }
    b headline --steps 60 --warmup 20
    b cfgH2g --steps 200 --warmup 50 --epochs 6553
    b unpruned --steps 60 --warmup 20 --variant 2
    b e2048 --steps 60 --warmup 20 --fft 2048
    b e1024 --steps 60 --warmup 20 --fft 1024
    b e512 --steps 60 --warmup 20 --fft 512
    b ref512 --steps 60 --warmup 20 --mode ref
    b welch --steps 40 --warmup 15 --mode welch
  done
done
python - <<PY
import json, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.load(open(f)); r = d['roofline']
        acc[f.split('/')[-1][:-5].rsplit('_', 1)[0]].append(r['frac'])
    except Exception as e:
        print(f, "ERR", open(f.replace('.json', '.err')).read()[-300:])
for tag in sorted({t.rsplit('_', 1)[0] for t in acc}):
    o, n = acc.get(tag + '_old', []), acc.get(tag + '_new', [])
    if o and n:
        mo, mn = sum(o) / len(o), sum(n) / len(n)
        print("%-10s old %s = %.4f   new %s = %.4f   new/old %+.2f %%" % (tag, " ".join("%.4f" % x for x in o), mo, " ".join("%.4f" % x for x in n), mn, 100 * (mn / mo - 1)))
PY
