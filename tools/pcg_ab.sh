# A/B of tools/variants/*.so on the PCG block of bench.py (K = 50 photometric)
for so in tools/variants/*.so; do
  BSLAM_HIP_LIB=$PWD/$so python bench.py ${PCG_AB_ARGS:---keyframes 50 --photometric 1} --secondary 0 --cpu-baseline 0 --steps 2 --warmup 1 > gpurun_out/pcg_ab.json
  python -c "
import json;d=json.load(open('gpurun_out/pcg_ab.json'));p=d['pcg']['headline_stack'];print('$so','ms/BA it',round(p['ms_per_ba_iteration'],2),'step1 us',round(p['pcg_step1_kernel']['avg_launch_us'],1),'init us',round(p['pcg_init_kernel']['avg_launch_us'],1),'pose us',round(d['roofline']['avg_launch_us'],1))"
done
