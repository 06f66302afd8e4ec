# Round 4: kernel trace of the training step (Trainer.step x 6, batch 512), every kernel listed
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_train_kt -- python3 tools/train_prof.py > gpurun_out/r04_train_kt.log 2>&1
echo rc=$?
python3 tools/top_kernels.py gpurun_out/r04_train_kt 60 > gpurun_out/r04_train_top.md 2>&1
find gpurun_out/r04_train_kt -name "*.csv" -size +3M -delete
tail -3 gpurun_out/r04_train_kt.log
cut -c1-200 gpurun_out/r04_train_top.md
