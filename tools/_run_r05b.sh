mkdir -p gpurun_out/r05_d
for n in 32 64 256 512 1024 4096; do for reo in 1 0; do
  ch=$(( n <= 1024 ? 5242 : (536870912 / n / 100) ))
  SMFFT_SCHEDULE_TRACE=/tmp/trace.txt python tools/workgroup_trace.py $n $ch -1 gpurun_out/r05_d/trace_${n}_${reo}.txt 100 $reo >> gpurun_out/r05_d/summary.txt 2>&1
  python tools/trace_summary.py gpurun_out/r05_d/trace_${n}_${reo}.txt >> gpurun_out/r05_d/summary.txt 2>&1
done; done
cat gpurun_out/r05_d/summary.txt
