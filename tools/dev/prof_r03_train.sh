# Round 3: kernel trace of the training step (Trainer.step x 12, batch 512)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_train_kt -- python3 tools/train_prof.py > gpurun_out/r03_train_kt.log 2>&1
echo rc=$?
python3 tools/top_kernels.py gpurun_out/r03_train_kt 30 > gpurun_out/r03_train_top.md 2>&1
find gpurun_out/r03_train_kt -name "*.csv" -size +3M -delete
tail -5 gpurun_out/r03_train_kt.log
