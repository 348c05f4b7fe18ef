# development aid (GPU box), round 4: allocator repro with its control, the run-length pre-pass as one launch against three, wide chunks cold / warm
mkdir -p gpurun_out/r4c
tools/mallocasync_repro.bin 2000 1 > gpurun_out/r4c/mallocasync.txt 2>&1
tools/mallocasync_repro.bin 4000 4 >> gpurun_out/r4c/mallocasync.txt 2>&1
tools/mallocasync_repro.bin 4000 2 >> gpurun_out/r4c/mallocasync.txt 2>&1
cat gpurun_out/r4c/mallocasync.txt
TAG=one_launch python tools/fine_prepass_sweep.py 28 > gpurun_out/r4c/fine_one.txt 2>/dev/null
GIDDY_RUN_FINE_ONE_LAUNCH=0 TAG=three_launches python tools/fine_prepass_sweep.py 28 > gpurun_out/r4c/fine_three.txt 2>/dev/null
paste -d'\n' gpurun_out/r4c/fine_one.txt gpurun_out/r4c/fine_three.txt
for v in default wide2 wide4; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for w in run_length_encoding run_position_encoding_relative run_position_encoding_absolute; do
    python tools/cold_input_probe.py $w 30 2>/dev/null | sed "s/^/$v /" | tee -a gpurun_out/r4c/cold_wide.txt
  done
done
