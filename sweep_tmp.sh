mkdir -p gpurun_out
(timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --size 128 --steps 2 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -12) > gpurun_out/bench_n2.log 2>&1
cat gpurun_out/bench_n2.log | cut -c1-1200
