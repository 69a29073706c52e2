#!/bin/bash
# rehearsal of bench.py's N > 1 path on a one-GPU box: N ranks share GPU 0 through the host-synchronous RCCL stand-in
# (tests/mock_rccl) -- checks the launch contract and the JSON line, not performance
N=${1:-2}
make -s -C tests/mock_rccl 2>/dev/null || (cd tests/mock_rccl && /opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -fPIC -shared mock_rccl.cpp -o libmock_rccl.so -lrt)
CWR_RCCL_LIB=$PWD/tests/mock_rccl/libmock_rccl.so CWR_BENCH_DEVICE=0 timeout -k 10 500 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 4 --warmup 2 --nx 400 --ny 400
