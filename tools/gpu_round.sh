#!/bin/bash
# Round evidence: GPU tests, smoke, headline bench, rocprofv3 kernel stats of the same command,
# HBM traffic counters (separate --pmc pass), side configs, host-buffer (PCIe-inclusive) rate.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r01
mkdir -p $O
cd $R
CRN_EVIDENCE_DIR=$O timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench_headline.json 2> $O/bench_headline.err
timeout 300 python bench.py --fft 1024 > $O/bench_cfg1_1024.json 2> $O/bench_cfg1.err
timeout 300 python bench.py --mode ref --cpu-epochs 0 > $O/bench_cfg3_ref512.json 2> $O/bench_cfg3.err
timeout 300 python bench.py --mode welch --cpu-epochs 0 > $O/bench_cfg2_welch.json 2> $O/bench_cfg2.err
timeout 300 python bench.py --variant 2 --cpu-epochs 0 > $O/bench_unpruned.json 2> $O/bench_unpruned.err
timeout 300 python bench.py --variant 16 --cpu-epochs 0 --no-check > $O/bench_noclose.json 2> $O/bench_noclose.err
timeout 300 python bench.py --fft 512 --cpu-epochs 0 > $O/bench_e512.json 2> $O/bench_e512.err
timeout 300 python bench.py --fft 2048 --cpu-epochs 0 > $O/bench_e2048.json 2> $O/bench_e2048.err
./tools/membw_policy > $O/membw_policy.txt 2>&1
timeout 300 python tools/host_rate.py > $O/host_rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --cpu-epochs 0 > $O/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 20 --cpu-epochs 0 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 5 --warmup 20 --cpu-epochs 0 > $O/pmc_write.log 2>&1
# HBM traffic of the side configurations (FETCH_SIZE and WRITE_SIZE in separate passes)
for cfgname in "cfg1:--fft 1024" "cfg3:--mode ref" "cfg2:--mode welch" "e512:--fft 512" "e2048:--fft 2048" "unpruned:--variant 2"; do
  tag=${cfgname%%:*}; args=${cfgname#*:}
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$tag -- python3 $R/bench.py --steps 5 --warmup 20 --cpu-epochs 0 $args > $O/pmc_fetch_$tag.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$tag -- python3 $R/bench.py --steps 5 --warmup 20 --cpu-epochs 0 $args > $O/pmc_write_$tag.log 2>&1
done
cd $R
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/bench_headline.json; cat $O/host_rate.txt
find $O/stats -name "*stats*.csv" | head
