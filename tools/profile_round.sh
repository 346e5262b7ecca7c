set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py > gpurun_out/bench_r1k_geo.json 2> gpurun_out/bench_r1k_geo.err
python bench.py --photometric 1 --cpu-baseline 0 > gpurun_out/bench_r1k_photo.json 2>/dev/null
python bench.py --keyframes 200 --cpu-baseline 0 > gpurun_out/bench_r1k_k200.json 2>/dev/null
python tools/bench_pcg.py > gpurun_out/bench_r1k_pcg.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1k_geo -- python3 $R/bench.py --cpu-baseline 0 > $R/gpurun_out/prof_r1k_geo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1k_photo -- python3 $R/bench.py --cpu-baseline 0 --photometric 1 > $R/gpurun_out/prof_r1k_photo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1k_pcg -- python3 $R/tools/bench_pcg.py > $R/gpurun_out/prof_r1k_pcg.log 2>&1
cd $R
bash tools/pmc.sh pmc_r1k_geo --steps 2 --warmup 1 > gpurun_out/pmc_r1k_geo.log 2>&1
bash tools/pmc.sh pmc_r1k_photo --steps 2 --warmup 1 --photometric 1 > gpurun_out/pmc_r1k_photo.log 2>&1
tail -c 400 gpurun_out/bench_r1k_geo.json
