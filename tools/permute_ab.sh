# tuning: surfel columns in creation order vs per-surfel Morton order (3-D / 2-D keys)
for mode in "" 3 2; do
  for ph in 0 1; do
    BSLAM_BENCH_PERMUTE=$mode python bench.py --keyframes 50 --photometric $ph --steps 5 --secondary 0 --pcg 0 --cpu-baseline 0 > gpurun_out/perm.json
    python -c "
import json;d=json.load(open('gpurun_out/perm.json'));r=d['roofline'];print('permute=[$mode] photometric=$ph ms/step',round(d['ms_per_step'],3),'pose us',round(r['avg_launch_us'],1),'frac',round(r['frac'],3),'geometry us',round(r['geometry_kernel']['us_per_step'],1),'frac',round(r['geometry_kernel']['frac'],3),'act us',round(r['activation_kernel']['avg_launch_us'],1))"
  done
done
