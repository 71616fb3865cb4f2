#!/bin/bash
# same-box A/B of two builds of libcrnsense.so (ab/libcrnsense_old.so vs ab/libcrnsense_new.so):
# boxes differ by ~5 %, so only numbers from one call are comparable.
mkdir -p gpurun_out/ab
run() {  # tag, lib, bench args...
  local tag=$1 lib=$2; shift 2
  CRN_SENSE_LIB=$PWD/ab/libcrnsense_$lib.so timeout 300 python bench.py --steps 20 --warmup 3 --cpu-epochs 0 "$@" \
      > gpurun_out/ab/${tag}_$lib.json 2> gpurun_out/ab/${tag}_$lib.err
}
for rep in 1 2; do
  for lib in old new; do
    run e4096_r$rep $lib
    run ref512_r$rep $lib --mode ref --fft 512
    run e1024_r$rep $lib --fft 1024
    run e512_r$rep $lib --fft 512
    run e2048_r$rep $lib --fft 2048
    run welch_r$rep $lib --mode welch
    run e4096v2_r$rep $lib --variant 2
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print("%-28s frac=%.4f kern_ms=%.4f value=%.1f"%(f.split('/')[-1][:-5], r['frac'], r['kernel_ms_mean'], d['value']))
    except Exception as e:
        print(f,"ERR",open(f.replace('.json','.err')).read()[-300:])
PY
