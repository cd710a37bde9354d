cd $GRAFT_REPO_ROOT
echo "== N=1 default, small scale"; timeout -k 10 300 python3 bench.py --scale 0.25 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | cut -c1-3000 || exit 1
echo "== N=2 self-launched over gloo (two ranks share the GPU)"; DISCO_DIST_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --scale 0.2 --steps 2 --warmup 1 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | cut -c1-4000 || exit 1
echo "== N=2 nccl must refuse on a 1-GPU box"; timeout -k 10 100 python3 bench.py --gpus 2 --steps 1; echo "rc=$?"
