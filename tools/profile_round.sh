# One GPU call: the evidence files of a round (copied into profiles/ afterwards, see profiles/README.md).
# usage (on the GPU box): bash tools/profile_round.sh <tag>      e.g. r02a
set -e
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"; tail -c 300 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
HEAD="--cpu-baseline 0 --secondary 0 --pcg 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k300_photo -- python3 $R/bench.py $HEAD > $O/stats_k300_photo.log 2>&1
echo "stats headline done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k50_geo -- python3 $R/bench.py $HEAD --keyframes 50 --photometric 0 > $O/stats_k50_geo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_k50_photo -- python3 $R/bench.py $HEAD --keyframes 50 --photometric 1 > $O/stats_k50_photo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pcg -- python3 $R/bench.py --cpu-baseline 0 --secondary 0 --pcg 1 --steps 1 --warmup 1 > $O/stats_pcg.log 2>&1
echo "stats done"
cd $R
bash tools/pmc.sh $TAG/pmc_k300_photo --steps 1 --warmup 1 --secondary 0 --pcg 0 > $O/pmc_k300_photo.log 2>&1
echo "pmc headline done"
bash tools/pmc.sh $TAG/pmc_k50_geo --steps 2 --warmup 1 --secondary 0 --pcg 0 --keyframes 50 --photometric 0 > $O/pmc_k50_geo.log 2>&1
bash tools/pmc.sh $TAG/pmc_k50_photo --steps 2 --warmup 1 --secondary 0 --pcg 0 --keyframes 50 --photometric 1 > $O/pmc_k50_photo.log 2>&1
echo "pmc done"
for w in k300_photo k50_geo k50_photo; do
  S=$(ls $O/stats_$w/*/*kernel_stats.csv | head -1)
  python tools/pmc_summary.py $O/pmc_$w $S > $O/pmc_summary_$w.txt 2>&1 || true
done
# keep only the summaries (the per-dispatch traces are large)
for d in $O/stats_*/; do n=$(basename $d); cp $(ls $d*/*kernel_stats.csv | head -1) $O/$n.kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
