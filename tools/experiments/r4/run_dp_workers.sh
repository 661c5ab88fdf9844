# two dp_worker.py ranks sharing cuda:0 (what tests/test_gpu_dp.py starts), with their full JSON lines
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 python3 tests/dp_worker.py > $1/dp_rank1.log 2>&1 &
P1=$!
RANK=0 timeout -k 10 300 python3 tests/dp_worker.py > $1/dp_rank0.log 2>&1
wait $P1
grep "^{" $1/dp_rank0.log $1/dp_rank1.log
