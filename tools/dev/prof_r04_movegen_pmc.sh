# Round 4 (final tree): the headline command under rocprofv3 -- kernel trace, then separate PMC passes (VALU / SALU, FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MG="--steps 200 --warmup 20 --windows 5 --no-cpu-baseline --no-overlap --selfplay-plies 0 --no-whole-games --train-steps 0 --encode-boards 0 --sat-boards 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_mg_kt -- python3 bench.py $MG > gpurun_out/r04_mg_kt.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r04_mg_valu -- python3 bench.py $MG > gpurun_out/r04_mg_valu.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r04_mg_f -- python3 bench.py $MG > gpurun_out/r04_mg_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04_mg_w -- python3 bench.py $MG > gpurun_out/r04_mg_w.log 2>&1
echo rc=$?
for d in r04_mg_kt r04_mg_valu r04_mg_f r04_mg_w; do python3 tools/prof_summary.py gpurun_out/$d hive_piece > gpurun_out/$d.md 2>&1; echo "== $d"; cut -c1-200 gpurun_out/$d.md | head -12; done
find gpurun_out/r04_mg_* -name "*.csv" -size +3M -delete
