for v in r1 r2 r4; do echo -n "$v: "; BSLAM_HIP_LIB=$PWD/tools/variants/libbadslam_hip_$v.so python tools/bench_pcg.py 2>&1 | grep "BA iteration"; done
for v in r1 r2 r4; do echo -n "$v photo: "; BSLAM_HIP_LIB=$PWD/tools/variants/libbadslam_hip_$v.so python tools/bench_pcg.py --photometric 1 2>&1 | grep "BA iteration"; done
