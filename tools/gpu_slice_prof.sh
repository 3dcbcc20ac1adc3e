set -u
R=$PWD
mkdir -p gpurun_out/slice
timeout -k 10 200 bash tools/gpu_check_new.sh "surface or rk2d or everything_but or 2d_loops" || exit 1
timeout -k 10 200 python -u tools/atm_slice_time.py 0 3 > gpurun_out/slice/atm_slice_time.txt 2>&1 || { tail -20 gpurun_out/slice/atm_slice_time.txt; exit 1; }
cat gpurun_out/slice/atm_slice_time.txt
timeout -k 10 200 python -u tools/hyd_slice_time.py > gpurun_out/slice/hyd_slice_time.txt 2>&1 || { tail -20 gpurun_out/slice/hyd_slice_time.txt; exit 1; }
cat gpurun_out/slice/hyd_slice_time.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/slice/prof -o atm -- python3 $R/tools/atm_slice_time.py 0 2 > $R/gpurun_out/slice/rocprof.log 2>&1 || { tail -20 $R/gpurun_out/slice/rocprof.log; exit 1; }
head -40 $R/gpurun_out/slice/prof/atm_kernel_stats.csv
