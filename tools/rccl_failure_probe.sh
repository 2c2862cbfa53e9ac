cd $GRAFT_REPO_ROOT
export SMFFT_BENCH_DEVICE=0 SMFFT_BENCH_PREWARM_S=0.2
echo "== fallback"; ( time timeout 700 python bench.py --gpus 2 --steps 3 --warmup 1 --nffts 65536 --no-cpu-baseline --no-configs > gpurun_out/rccl_fb.out 2> gpurun_out/rccl_fb.err ); echo rc=$?; tail -c 600 gpurun_out/rccl_fb.out; grep -i "bench\]\|error\|duplicate" gpurun_out/rccl_fb.err | head -8
echo "== require"; ( time SMFFT_BENCH_REQUIRE_RCCL=1 timeout 700 python bench.py --gpus 2 --steps 3 --warmup 1 --nffts 65536 --no-cpu-baseline --no-configs > gpurun_out/rccl_rq.out 2> gpurun_out/rccl_rq.err ); echo rc=$?; grep -i "bench\]" gpurun_out/rccl_rq.err | head -4
