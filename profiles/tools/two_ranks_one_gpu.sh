cd $GRAFT_REPO_ROOT
export AL_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 python bench.py --gpus 2 --config cfg1 --steps 20 --warmup 3 > gpurun_out/two_rank1.txt 2>&1 &
RANK=0 python bench.py --gpus 2 --config cfg1 --steps 20 --warmup 3 > gpurun_out/two_rank0.txt 2>&1
wait
