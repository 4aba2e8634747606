#!/bin/bash
# usage (GPU box): scripts/refresh_profiles.sh [round-tag]  -- the bench lines, probe, counter and test summaries kept under profiles/
# (written to gpurun_out/refresh/, copied into profiles/ by hand after a look).  Every command runs under a timeout: a kernel that
# hangs must not hold the box.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
{
  echo '$ python -m pytest tests/ -x -q -m gpu'
  timeout 600 python -m pytest tests/ -x -q -m gpu --timeout 200 -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3
  echo '$ python -m pytest (files in reverse order) -x -q -m gpu'
  timeout 600 python -m pytest $(ls tests/test_gpu_*.py | sort -r) -x -q -m gpu --timeout 200 -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3
  echo '$ python -m pytest tests/test_gpu_tasks.py -x -q -m gpu'
  timeout 600 python -m pytest tests/test_gpu_tasks.py -x -q -m gpu --timeout 200 -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror" | tail -3
  echo '$ python scripts/fuzz_parity.py 1000 600000'
  timeout 900 python scripts/fuzz_parity.py 1000 600000 2>&1 | tail -3
} > $O/${TAG}_gputest_final.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_webbase-1M.json 2> $O/bench_webbase.err
timeout 300 python bench.py --two-phase --no-cpu-baseline > $O/${TAG}_bench_webbase-1M_two_phase.json 2>> $O/bench_webbase.err
for w in cop20k_A cage12 mc2depi rmat16 rmat18; do
    timeout 400 python bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2> $O/bench_$w.err
done
timeout 300 python bench.py --accumulator sort_merge --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_webbase-1M_sort_merge.json 2> $O/bench_sm.err
timeout 300 python bench.py --accumulator sort_merge --workload cop20k_A --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cop20k_A_sort_merge.json 2>> $O/bench_sm.err
timeout 300 python scripts/probe_tasks.py webbase cop20k cage12 mc2depi rmat16 rmat18 > $O/${TAG}_probe_tasks.txt 2>&1
timeout 300 python scripts/probe_blocks.py > $O/${TAG}_probe_blocks.txt 2>&1
timeout 300 python scripts/probe_floor.py webbase cop20k > $O/${TAG}_floor_raw.txt 2>&1
timeout 300 python scripts/class_costs.py webbase > $O/${TAG}_class_costs.txt 2>/dev/null
for w in webbase-1M cop20k_A; do
    timeout 900 scripts/collect_traffic.sh $w $TAG > $O/collect_$w.log 2>&1
    cp $R/gpurun_out/traffic_$w/${TAG}_* $O/ 2>/dev/null
done
timeout 700 scripts/collect_sq.sh webbase-1M ${TAG}_sq_webbase-1M > $O/collect_sq.log 2>&1
timeout 700 scripts/collect_sq.sh webbase-1M ${TAG}_sq_webbase-1M_two_phase --two-phase >> $O/collect_sq.log 2>&1
cp $R/gpurun_out/sq_${TAG}_sq_webbase-1M/${TAG}_sq_webbase-1M.json $R/gpurun_out/sq_${TAG}_sq_webbase-1M_two_phase/${TAG}_sq_webbase-1M_two_phase.json $O/ 2>/dev/null
timeout 600 python bench.py --workload rmat22 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_rmat22_1gpu.json 2> $O/bench_rmat22.err
timeout 600 python bench.py --workload rmat22 --steps 2 --warmup 1 --no-cpu-baseline --chunk-consumer checksum > $O/${TAG}_bench_rmat22_1gpu_checksum.json 2>> $O/bench_rmat22.err
ls -la $O
