#!/bin/bash
# usage (GPU box): scripts/refresh_profiles.sh [round-tag]  -- the bench lines, probe and traffic summaries kept under profiles/
# (written to gpurun_out/refresh/, copied into profiles/ by hand after a look)
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_webbase-1M.json 2> $O/bench_webbase.err
python bench.py --two-phase --no-cpu-baseline > $O/${TAG}_bench_webbase-1M_two_phase.json 2>> $O/bench_webbase.err
for w in cop20k_A cage12 mc2depi rmat16 rmat18; do
    python bench.py --workload $w --no-cpu-baseline > $O/${TAG}_bench_$w.json 2> $O/bench_$w.err
done
python bench.py --accumulator sort_merge --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_webbase-1M_sort_merge.json 2> $O/bench_sm.err
python bench.py --accumulator sort_merge --workload cop20k_A --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_cop20k_A_sort_merge.json 2>> $O/bench_sm.err
python scripts/probe_tasks.py webbase cop20k cage12 mc2depi rmat16 rmat18 > $O/${TAG}_probe_tasks.txt 2>&1
python scripts/probe_blocks.py > $O/${TAG}_probe_blocks.txt 2>&1
python scripts/class_costs.py webbase > $O/${TAG}_class_costs.txt 2>/dev/null
for w in webbase-1M cop20k_A; do
    timeout 600 scripts/collect_traffic.sh $w $TAG > $O/collect_$w.log 2>&1
    cp $R/gpurun_out/traffic_$w/${TAG}_* $O/ 2>/dev/null
done
timeout 600 scripts/collect_sq.sh webbase-1M ${TAG}_sq_webbase-1M > $O/collect_sq.log 2>&1
timeout 600 scripts/collect_sq.sh webbase-1M ${TAG}_sq_webbase-1M_two_phase --two-phase >> $O/collect_sq.log 2>&1
cp $R/gpurun_out/sq_${TAG}_sq_webbase-1M/${TAG}_sq_webbase-1M.json $R/gpurun_out/sq_${TAG}_sq_webbase-1M_two_phase/${TAG}_sq_webbase-1M_two_phase.json $O/ 2>/dev/null
ls -la $O
