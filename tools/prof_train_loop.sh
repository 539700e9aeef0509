# On the GPU box: kernel timeline of ONE batched step of the device-resident learning loop (S3 env step + replay + optimiser chain)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trl -- python3 $R/tools/time_train_device.py 128 30 > $R/gpurun_out/trl.log 2>&1
tail -4 $R/gpurun_out/trl.log
python3 $R/tools/timeline_step.py $R/gpurun_out/trl smooth_linear_kernel -3 > $R/gpurun_out/trl_timeline.txt 2>&1
cat $R/gpurun_out/trl_timeline.txt
find $R/gpurun_out/trl -name "*kernel_trace.csv" -delete
