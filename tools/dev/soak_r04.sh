# Round 4: the full-size replays of DESIGN section 4 on the final tree (default fp16 engine, balanced 72-tile tower)
cd $GRAFT_REPO_ROOT
(while true; do date >> gpurun_out/r04_soak_heartbeat.log; sleep 50; done) &
HB=$!
trap "kill $HB" EXIT
HIVE_SOAK_GAMES=1024 timeout -k 10 500 python -m pytest tests/test_gpu_scale.py -x -q -s -k "finished_selfplay_games_replay" > gpurun_out/r04_soak_replay1024.log 2>&1 &&
grep -h "oracle replay\|passed\|failed" gpurun_out/r04_soak_replay1024.log &&
HIVE_SOAK_GAMES=256 HIVE_SOAK_SIMS=250 HIVE_SOAK_SLOTS=4 timeout -k 10 500 python -m pytest tests/test_gpu_scale.py -x -q -s -k "finished_selfplay_games_replay" > gpurun_out/r04_soak_250x4.log 2>&1 &&
grep -h "oracle replay\|passed\|failed" gpurun_out/r04_soak_250x4.log &&
HIVE_TEST_HEAVY=1 timeout -k 10 300 python -m pytest tests/test_gpu_env.py -x -q -k "lockstep" > gpurun_out/r04_soak_lockstep.log 2>&1 &&
tail -1 gpurun_out/r04_soak_lockstep.log &&
HIVE_SOAK_WORKER_GAMES=16384 HIVE_SOAK_SIMS=50 HIVE_SOAK_PER_GPU=1024 HIVE_SOAK_REPLAY=512 timeout -k 10 1000 python -m pytest tests/test_gpu_scale.py -x -q -s -k "worker_files_replay" > gpurun_out/r04_soak_worker.log 2>&1 &&
grep -h "worker files\|passed\|failed" gpurun_out/r04_soak_worker.log
