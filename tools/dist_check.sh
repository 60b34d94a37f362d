# dev tool: single-rank RCCL path vs plain single-process step (GPU box)
mkdir -p gpurun_out
a=$(python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
b=$(AVID_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
echo "plain: $a   1-rank RCCL: $b" | tee -a gpurun_out/dist_check.txt
