# One GPU call: the evidence files of a round (copied into profiles/ afterwards, see profiles/README.md).
# usage (on the GPU box): bash tools/profile_round.sh <tag> [bench|stats|pmc|pmc2|post]      e.g. r03a bench
# (one stage per call keeps a call under the pool's 20-minute limit; no stage = all of them)
set -e
TAG=${1:-r03a}
STAGE=${2:-all}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
ONE="--cpu-baseline 0 --secondary 0 --pcg 0 --trajectory 0 --survey 0 --strong 0"
if [ $STAGE = all ] || [ $STAGE = bench ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; tail -c 300 $O/bench_default.json
python bench.py $ONE --keyframes 200 --photometric 0 > $O/bench_k200_geo.json 2> $O/bench_k200_geo.err
python bench.py $ONE --keyframes 1000 --photometric 0 --steps 3 --warmup 1 > $O/bench_k1000_geo.json 2> $O/bench_k1000_geo.err
python bench.py $ONE --keyframes 50 --photometric 1 > $O/bench_k50_photo.json 2> $O/bench_k50_photo.err
# BASELINE.json configs[4]'s regime on the smooth-trajectory stack ("a smooth 1000-pose trajectory for C5")
python bench.py $ONE --scene trajectory --keyframes 1000 --photometric 0 --steps 3 --warmup 1 > $O/bench_k1000_trajectory_geo.json 2> $O/bench_k1000_trajectory_geo.err
echo "extra bench lines done"
fi
if [ $STAGE = all ] || [ $STAGE = stats ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k300_photo -- python3 $R/bench.py $ONE > $O/stats_k300_photo.log 2>&1
echo "stats headline done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k300_trajectory -- python3 $R/bench.py $ONE --scene trajectory --steps 5 > $O/stats_k300_trajectory.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k50_geo -- python3 $R/bench.py $ONE --keyframes 50 --photometric 0 > $O/stats_k50_geo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k50_photo -- python3 $R/bench.py $ONE --keyframes 50 --photometric 1 > $O/stats_k50_photo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pcg -- python3 $R/bench.py --cpu-baseline 0 --secondary 0 --pcg 1 --trajectory 0 --survey 0 --strong 0 --steps 1 --warmup 1 > $O/stats_pcg.log 2>&1
echo "stats done"
for d in $O/stats_*/; do n=$(basename $d); cp $(ls $d*/*kernel_stats.csv | head -1) $O/$n.kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete
fi
cd $R
if [ $STAGE = all ] || [ $STAGE = pmc ]; then
bash tools/pmc.sh $TAG/pmc_k300_photo --steps 1 --warmup 1 --secondary 0 --pcg 1 --trajectory 0 --survey 0 --strong 0 > $O/pmc_k300_photo.log 2>&1
echo "pmc headline done"
fi
if [ $STAGE = all ] || [ $STAGE = pmc2 ]; then
bash tools/pmc.sh $TAG/pmc_k300_trajectory --steps 2 --warmup 1 --secondary 0 --pcg 0 --trajectory 0 --survey 0 --strong 0 --scene trajectory > $O/pmc_k300_trajectory.log 2>&1
bash tools/pmc.sh $TAG/pmc_k50_geo --steps 2 --warmup 1 --secondary 0 --pcg 1 --trajectory 0 --survey 0 --strong 0 --keyframes 50 --photometric 0 > $O/pmc_k50_geo.log 2>&1
bash tools/pmc.sh $TAG/pmc_k50_photo --steps 2 --warmup 1 --secondary 0 --pcg 0 --trajectory 0 --survey 0 --strong 0 --keyframes 50 --photometric 1 > $O/pmc_k50_photo.log 2>&1
echo "pmc done"
fi
# summaries of whatever PMC passes this call produced (kernel durations: the stats stage's summary, from this call or -- stages run
# in separate GPU calls -- from the copy committed under profiles/)
for w in k300_photo k300_trajectory k50_geo k50_photo; do
  [ -d $O/pmc_$w ] || continue
  # (a stage run earlier in the same session has had its counter files summarised and deleted: leave its summaries alone)
  [ -n "$(find $O/pmc_$w -name '*counter_collection.csv' | head -1)" ] || continue
  S=$(ls $O/stats_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$S" ] || S=$R/profiles/${TAG}_kernel_stats_$w.csv
  python tools/pmc_summary.py $O/pmc_$w $S > $O/pmc_summary_$w.txt 2>&1 || true
  python tools/pmc_traffic.py $O/pmc_$w "profiles/${TAG}_pmc_summary_$w.txt + the counter CSVs of the same tools/pmc.sh run" > $O/pmc_traffic_$w.json 2> $O/pmc_traffic_$w.err || true
done
# keep only the summaries (the per-dispatch counter files are large)
find $O -name "*agent_info.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O
