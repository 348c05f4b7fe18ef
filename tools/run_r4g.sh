# development aid (GPU box), round 4: anchors touched ahead (cold / warm), the size probe for the fixed-rate decoders, the slowest GPU tests
mkdir -p gpurun_out/r4g
for v in default touch512 touch2048 touch8192 default touch2048; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for w in run_length_encoding run_position_encoding_relative run_position_encoding_absolute; do
    python tools/cold_input_probe.py $w 30 2>/dev/null | sed -e "s/, 'note'.*//" -e "s/^/$v /" | tee -a gpurun_out/r4g/cold_touch.txt
  done
done
unset GIDDY_LIBRARY_PATH
for w in dictionary frame_of_reference delta_patched run_length_encoding; do python tools/size_probe.py $w 30 4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4g/size_probe.txt; done
python tools/size_probe.py dictionary 32 4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4g/size_probe.txt
python tools/size_probe.py discard_zero_bytes_variable 31 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4g/size_probe.txt
python tools/mixed_density.py 28 2>/dev/null | tee gpurun_out/r4g/mixed_density.txt
python tools/mixed_density.py 30 2>/dev/null | tee -a gpurun_out/r4g/mixed_density.txt
python tools/compress_bench.py 28 2>/dev/null | tee gpurun_out/r4g/compress_bench.txt
timeout 1500 python -m pytest tests -q -m gpu --durations=12 -x 2>&1 | tail -20 | tee gpurun_out/r4g/durations.txt
