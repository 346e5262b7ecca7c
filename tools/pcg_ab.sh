# A/B of tools/variants/*.so on the PCG block of bench.py.  The PCG block runs through the C++ host class, whose library links
# badslam_amd/libbadslam_hip.so by path: each variant is therefore copied over that file (on the GPU box's scratch copy).
cp badslam_amd/libbadslam_hip.so /tmp/libbadslam_hip_main.so
for so in tools/variants/*.so; do
  cp $so badslam_amd/libbadslam_hip.so
  python bench.py ${PCG_AB_ARGS:---keyframes 50 --photometric 1} --secondary 0 --cpu-baseline 0 --steps 2 --warmup 1 > gpurun_out/pcg_ab.json
  python -c "
import json;d=json.load(open('gpurun_out/pcg_ab.json'));p=d['pcg']['headline_stack'];print('$so','ms/BA it',round(p['ms_per_ba_iteration'],2),'step1 us',round(p['pcg_step1_kernel']['avg_launch_us'],1),'init us',round(p['pcg_init_kernel']['avg_launch_us'],1),'pose us',round(d['roofline']['avg_launch_us'],1))"
done
cp /tmp/libbadslam_hip_main.so badslam_amd/libbadslam_hip.so
