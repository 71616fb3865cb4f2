cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in old new; do
  export CRN_SENSE_LIB=$R/ab/libcrnsense_$lib.so
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pw_$lib -- python3 $R/bench.py --mode welch --steps 3 --warmup 3 --cpu-epochs 0 > $R/gpurun_out/pw_$lib.log 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$R/gpurun_out/pw_$lib/*/*_counter_collection.csv"))[-1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "sense_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("$lib", "FETCH_SIZE KiB", sum(v[-3:])/3, "x2 bytes", sum(v[-3:])/3*2048)
PY
done
