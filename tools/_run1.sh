cd $GRAFT_REPO_ROOT
python tools/dbg_ws.py 2 288 288 0 2>&1 | grep "^rep"
python tools/dbg_ws.py 2 288 288 250 2>&1 | grep "^rep"
python -m pytest tests/test_hip_gpu.py -q -x -k "akgm" 2>&1 | tail -3
bash tools/kprof.sh ws python $GRAFT_REPO_ROOT/tools/bench_op.py akgm 16 288 288 64 10 | grep -i "akgm"
UCDIR_NO_WS=1 bash tools/kprof.sh pre python $GRAFT_REPO_ROOT/tools/bench_op.py akgm 16 288 288 64 10 | grep -i akgm
