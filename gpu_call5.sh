O=gpurun_out/r06; mkdir -p $O
# the driver's command, on this commit
( time CRN_EVIDENCE_DIR=$PWD/$O timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
cp $O/roofline_floors.json $O/roofline_floors_green.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
( time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
timeout 600 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
# kernel stats of the headline command (the program itself after --)
cd /tmp && export TMPDIR=/tmp
PYTHON=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- $PYTHON $R/bench.py --cpu-epochs 0 --no-live-traffic --no-alt > $R/$O/stats.log 2>&1
cd $R
# the gate against an -O1 build of every kernel (plain hipcc; -O0 does not compile: backend error "illegal VGPR to SGPR copy")
(time make -B -j8 -C cognitive-radio-network_amd/csrc plain "FLAGS=--offload-arch=gfx950 -O1 -fno-slp-vectorize -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -pthread -D__HIP_PLATFORM_AMD__ -x hip --offload-compress") > $O/build_O1.log 2>&1
CRN_EVIDENCE_DIR=$PWD/$O CRN_SENSE_LIB=$PWD/cognitive-radio-network_amd/libcrnsense_plain.so timeout 1500 python -m pytest tests/test_zz_roofline_floors.py -m gpu -q > $O/floor_gate_O1.log 2>&1; echo "pytest exit $?" >> $O/floor_gate_O1.log
cp $O/roofline_floors.json $O/roofline_floors_O1.json
tail -6 $O/pytest_gpu.log; tail -2 $O/smoke.log; cut -c1-300 $O/bench_driver_shape.json; tail -4 $O/build_O1.log; tail -5 $O/floor_gate_O1.log; find $O/stats -name "*kernel_stats.csv" | head -2
