mkdir -p gpurun_out/r5s2n
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_driver.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3 > gpurun_out/r5s2n/test.txt
for i in 1 2; do
C2_LIB_PATH=$PWD/celerite2_amd/libcelerite2_amd_base.so C2_SCAN_FWD_ONLY=1 python tools/nrhs_scan.py 1 2 3 4 5 > gpurun_out/r5s2n/scan_base_$i.txt 2>&1
C2_SCAN_FWD_ONLY=1 python tools/nrhs_scan.py 1 2 3 4 5 > gpurun_out/r5s2n/scan_new_$i.txt 2>&1
done
tail -n 5 gpurun_out/r5s2n/test.txt gpurun_out/r5s2n/scan_*.txt
