# how long the two gloo ranks of tests/test_dist.py::test_two_ranks_compare_device_rows take, and where
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 WORLD_SIZE=2 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1 SPSP_TEST_ROWS=block
t0=$(date +%s.%N)
RANK=0 LOCAL_RANK=0 python3 -X importtime $R/tests/dist_worker.py gpu 2> /tmp/r0.err > /tmp/r0.out &
RANK=1 LOCAL_RANK=1 python3 $R/tests/dist_worker.py gpu > /tmp/r1.out 2>&1
wait
t1=$(date +%s.%N)
echo "two ranks: $(echo "$t1 - $t0" | bc) s"
sort -t'|' -k2 -n /tmp/r0.err | tail -8
