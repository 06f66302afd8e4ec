# Round 3, planes writer after the streaming-store change: kernel trace + FETCH_SIZE + WRITE_SIZE passes of tools/encode_bench.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_enc_kt -- python3 tools/encode_bench.py 65536 > gpurun_out/r03_enc_kt.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03_enc_f -- python3 tools/encode_bench.py 65536 > gpurun_out/r03_enc_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03_enc_w -- python3 tools/encode_bench.py 65536 > gpurun_out/r03_enc_w.log 2>&1
echo rc=$?
for d in r03_enc_kt r03_enc_f r03_enc_w; do python3 tools/prof_summary.py gpurun_out/$d expand > gpurun_out/$d.md 2>&1; cat gpurun_out/$d.md; done
find gpurun_out/r03_enc_* -name "*.csv" -size +1M -delete
