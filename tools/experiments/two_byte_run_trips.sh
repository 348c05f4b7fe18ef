# run decoders, 2-byte columns: one segment of 4096 outputs per trip (variants/before.so) against two (variants/two8192.so: -DGIDDY_RUN_TWO_BYTE_TRIP_ELEMENTS=8192), 2^30 outputs
for v in before two8192 before two8192; do
  export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so
  SWEEP_REPS=9 timeout 600 python tools/sweep_one.py "run_length_encoding::kernel_t<[48]u, 2u|run_position_encoding::kernel_t<[48]u, 2u" 30 2>&1 | grep "kernel_t" | sed -E "s/^/$v: /; s/ +/ /g; s/decompression:://"
done
